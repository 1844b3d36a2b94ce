"""Dataset and metadata registries with the surface of detectron2/data/catalog.py:18-236: `DatasetCatalog` maps a name to a
loader function, `MetadataCatalog` maps a name to a bag of attributes that may be set once and never changed."""


class _Catalog:
    """Name -> entry table shared by the two registries (membership test, listing, removal)."""

    def __init__(self, kind):
        self._kind, self._entries = kind, {}

    def __contains__(self, name):
        return name in self._entries

    def keys(self):
        return self._entries.keys()

    def list(self):
        return list(self._entries)

    def remove(self, name):
        del self._entries[name]

    pop = remove

    def clear(self):
        self._entries.clear()


class _DatasetCatalog(_Catalog):
    def __init__(self):
        super().__init__("dataset")

    def register(self, name, func):
        if not callable(func):
            raise AssertionError("DatasetCatalog.register needs a function returning the dataset dicts, got %r" % (func,))
        if name in self._entries:
            raise AssertionError("dataset '%s' is registered already" % name)
        self._entries[name] = func

    def get(self, name):
        if name not in self._entries:
            raise KeyError("dataset '%s' is not registered; known: %s" % (name, ", ".join(self._entries) or "none"))
        return self._entries[name]()


class Metadata:
    """Attribute bag.  An attribute can be assigned the same value any number of times; a different value is an error
    (catalog.py:139-151: two loaders disagreeing about, say, the class list is a bug, not something to paper over)."""

    def __init__(self, name):
        object.__setattr__(self, "_values", {"name": name})

    def __getattr__(self, key):
        values = object.__getattribute__(self, "_values")
        if key in values:
            return values[key]
        raise AttributeError("metadata of '%s' has no '%s' (it has: %s)" % (values["name"], key, ", ".join(values)))

    def __setattr__(self, key, value):
        values = object.__getattribute__(self, "_values")
        if key in values and values[key] != value:
            raise AssertionError("metadata '%s' of '%s' is already %r; refusing to replace it by %r"
                                 % (key, values["name"], values[key], value))
        values[key] = value

    def set(self, **kwargs):
        for key, value in kwargs.items():
            setattr(self, key, value)
        return self

    def get(self, key, default=None):
        return object.__getattribute__(self, "_values").get(key, default)

    def as_dict(self):
        return dict(object.__getattribute__(self, "_values"))


class _MetadataCatalog(_Catalog):
    def __init__(self):
        super().__init__("metadata")

    def get(self, name):
        if not name:
            raise AssertionError("metadata needs a dataset name")
        if name not in self._entries:
            self._entries[name] = Metadata(name)
        return self._entries[name]


DatasetCatalog = _DatasetCatalog()
MetadataCatalog = _MetadataCatalog()
