"""Dataset and metadata registries (detectron2/data/catalog.py:18-236): names -> loader functions, names -> attribute bags."""
import types


class _DatasetCatalog(dict):
    def register(self, name, func):
        assert callable(func), "You must register a function with `DatasetCatalog.register`!"
        assert name not in self, "Dataset '{}' is already registered!".format(name)
        self[name] = func

    def get(self, name):
        try:
            f = self[name]
        except KeyError as e:
            raise KeyError("Dataset '{}' is not registered! Available datasets are: {}".format(
                name, ", ".join(list(self.keys())))) from e
        return f()

    def list(self):
        return list(self.keys())

    def remove(self, name):
        self.pop(name)


class Metadata(types.SimpleNamespace):
    name = "N/A"

    def __setattr__(self, key, val):
        # the reference refuses to change a value once set (catalog.py:139-151): silent drift between loaders is a bug
        if key in self.__dict__ and self.__dict__[key] != val:
            raise AssertionError("Attribute '{}' in the metadata of '{}' cannot be set to a different value!\n{} != {}".format(
                key, self.name, self.__dict__[key], val))
        super().__setattr__(key, val)

    def __getattr__(self, key):
        raise AttributeError("Attribute '{}' does not exist in the metadata of dataset '{}'. Available keys are {}.".format(
            key, self.name, str(list(self.__dict__.keys()))))

    def as_dict(self):
        return dict(self.__dict__)

    def set(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
        return self

    def get(self, key, default=None):
        return self.__dict__.get(key, default)


class _MetadataCatalog(dict):
    def get(self, name):
        assert len(name)
        if name not in self:
            self[name] = Metadata(name=name)
        return self[name]

    def list(self):
        return list(self.keys())

    def remove(self, name):
        self.pop(name)


DatasetCatalog = _DatasetCatalog()
MetadataCatalog = _MetadataCatalog()
