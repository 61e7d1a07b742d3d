"""dgl.data.DGLDataset: constructor runs process() unless a cache exists, then save()."""
from . import utils  # noqa: F401


class DGLDataset:
    def __init__(self, name, url=None, raw_dir=None, save_dir=None, hash_key=(),
                 force_reload=False, verbose=False, transform=None):
        self._name = name
        self._force_reload = force_reload
        self._verbose = verbose
        self._load()

    def _load(self):
        if not self._force_reload and self.has_cache():
            self.load()
            return
        self.download()
        self.process()
        self.save()

    def download(self):
        pass

    def has_cache(self):
        return False

    def save(self):
        pass

    def load(self):
        pass

    @property
    def name(self):
        return self._name
