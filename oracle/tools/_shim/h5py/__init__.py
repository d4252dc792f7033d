"""In-memory stand-in for h5py (not installed in this image), used ONLY by oracle/tools/make_goldens.py to run the
reference's export_predictions and record which groups / datasets it writes.  No arithmetic."""
import numpy as np

FILES = {}


class Dataset:
    def __init__(self, data):
        self.data = np.asarray(data)

    def __array__(self, dtype=None):
        return self.data if dtype is None else self.data.astype(dtype)


class Group(dict):
    def create_group(self, name):
        if name in self:
            raise RuntimeError(f"group {name} exists")
        g = Group()
        self[name] = g
        return g

    def create_dataset(self, name, data=None):
        self[name] = Dataset(data)
        return self[name]


class File(Group):
    def __init__(self, path, mode="r"):
        super().__init__()
        if mode == "r":  # re-open what an earlier File(path, "w") recorded (CacheLoader reads it back)
            self.update(FILES[str(path)])
        else:
            FILES[str(path)] = self

    def close(self):
        pass
