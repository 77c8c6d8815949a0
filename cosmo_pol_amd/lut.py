"""Scattering lookup table container with the interface of lookup/lut.py:79-245
(`axes`, `axes_names`, `axes_limits`, `axes_step`, `value_table`, `lookup_line`).

The table itself is consumed on the GPU (uploaded once per class); lookup_line is kept for
API compatibility and for host-side inspection: nearest node, np.round (half to even) on
float32 axis arithmetic, clamped (lut.py:240-241).
"""
import numpy as np


class Error(Exception):
    pass


class Lookup_table(object):
    def __init__(self):
        self.axes_names = {}
        self.axes_limits = []
        self.axes_step = []
        self.axes = []
        self.value_table = []

    def add_axis(self, name, axis_values=None, is_regular=False):
        if name in self.axes_names:
            raise Error("Axis already exists with name: '%s'" % name)
        self.axes_names[name] = len(self.axes)
        v = np.asarray(axis_values).astype('float32')
        self.axes_limits.append([np.min(v), np.max(v)])
        self.axes_step.append(v[1] - v[0])
        self.axes.append(v)

    def set_value_table(self, value_table, map_to_int=False):
        if map_to_int:
            raise Error("int16-packed tables are not supported by the GPU path")
        self.value_table = value_table if isinstance(value_table, np.ndarray) else np.array(value_table)

    def node_index(self, name, x):
        ax = self.axes_names[name]
        c = np.round((np.asarray(x) - self.axes_limits[ax][0]) / self.axes_step[ax])
        return np.minimum(np.maximum(0, c), self.value_table.shape[ax] - 1).astype(int)

    def lookup_line(self, **kwargs):
        idx = [slice(None)] * self.value_table.ndim
        for name, x in kwargs.items():
            idx[self.axes_names[name]] = self.node_index(name, x)
        return self.value_table[tuple(idx)]

    @classmethod
    def from_arrays(cls, e_axis, t_axis, d_axis, table):
        lut = cls()
        lut.add_axis('e', e_axis)
        lut.add_axis('t', t_axis)
        lut.add_axis('d', d_axis)
        lut.add_axis('sz', np.arange(12))
        lut.set_value_table(np.ascontiguousarray(table, dtype=np.float32))
        return lut
