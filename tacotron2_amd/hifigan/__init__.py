"""Drop-in for the reference's `hifigan` package (hifigan/__init__.py:1-7, hifigan/models.py)."""
from .models import Generator  # noqa: F401


class AttrDict(dict):
    """JSON config with attribute access, as the drivers build it (do_syn.py:156-158)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self
