"""Instances: the per-image table the model passes between its parts.

API-compatible with detectron2/structures/instances.py (the ROI heads, the data mapper and the evaluators address it as
``x.gt_boxes``, ``x.has("gt_masks")``, ``x[mask]``, ``Instances.cat([...])``), organised here as a column store: an
ordered name -> column map plus the common row count, which is fixed by the first column and enforced for every later one.
A column is anything with ``len`` and row indexing (Tensor, Boxes, BitMasks, list)."""
import torch


def _concat_columns(cols):
    """Row-wise concatenation of same-typed columns."""
    head = cols[0]
    if torch.is_tensor(head):
        return torch.cat(cols, dim=0)
    if isinstance(head, (list, tuple)):
        out = []
        for c in cols:
            out.extend(c)
        return out
    joiner = getattr(type(head), "cat", None)
    if joiner is None:
        raise TypeError("Instances.cat: no way to concatenate columns of type %s" % type(head).__name__)
    return joiner(cols)


class Instances:
    __slots__ = ("_image_size", "_columns", "_rows", "__weakref__")  # weak references: PaddedTargets caches per batch

    def __init__(self, image_size, **columns):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_columns", {})
        object.__setattr__(self, "_rows", None)
        for name, col in columns.items():
            self.set(name, col)

    # ---- columns ------------------------------------------------------------------------------
    @property
    def image_size(self):
        """(height, width) of the image the rows live in."""
        return self._image_size

    def set(self, name, column):
        n = len(column)
        if self._rows is None or not self._columns:
            object.__setattr__(self, "_rows", n)
        elif n != self._rows:
            raise AssertionError("column %r has %d rows, the table has %d" % (name, n, self._rows))
        self._columns[name] = column

    def get(self, name):
        return self._columns[name]

    def has(self, name):
        return name in self._columns

    def remove(self, name):
        self._columns.pop(name)
        if not self._columns:
            object.__setattr__(self, "_rows", None)

    def get_fields(self):
        return self._columns

    def __setattr__(self, name, column):
        if name in ("_image_size", "_columns", "_rows"):
            object.__setattr__(self, name, column)
        else:
            self.set(name, column)

    def __getattr__(self, name):
        # only reached when normal lookup fails, i.e. for column names
        try:
            return object.__getattribute__(self, "_columns")[name]
        except KeyError:
            raise AttributeError("Instances has no column %r (columns: %s)" % (name, ", ".join(self._columns) or "none")) from None

    # ---- rows ---------------------------------------------------------------------------------
    def __len__(self):
        if self._rows is None:
            raise NotImplementedError("an Instances without columns has no length")
        return self._rows

    def __iter__(self):
        raise NotImplementedError("Instances is a table, not a sequence of rows: index it instead")

    def __getitem__(self, rows):
        """Row selection: int (kept as a one-row table), slice, index tensor or boolean mask."""
        if isinstance(rows, int):
            n = len(self)
            if not -n <= rows < n:
                raise IndexError("row %d of an Instances with %d rows" % (rows, n))
            rows = rows % n
            rows = slice(rows, rows + 1)
        picked = Instances(self._image_size)
        for name, col in self._columns.items():
            picked.set(name, col[rows])
        return picked

    def to(self, *args, **kwargs):
        moved = Instances(self._image_size)
        for name, col in self._columns.items():
            moved.set(name, col.to(*args, **kwargs) if hasattr(col, "to") else col)
        return moved

    @staticmethod
    def cat(tables):
        tables = list(tables)
        if not tables or not all(isinstance(t, Instances) for t in tables):
            raise AssertionError("Instances.cat needs a non-empty list of Instances")
        if len(tables) == 1:
            return tables[0]
        joined = Instances(tables[0].image_size)
        for name in tables[0]._columns:
            joined.set(name, _concat_columns([t.get(name) for t in tables]))
        return joined

    def __repr__(self):
        rows = "?" if self._rows is None else str(self._rows)
        cols = ", ".join("%s: %s" % (k, v) for k, v in self._columns.items())
        return "Instances(num_instances=%s, image_height=%s, image_width=%s, fields=[%s])" % (
            rows, self._image_size[0], self._image_size[1], cols)

    __str__ = __repr__
