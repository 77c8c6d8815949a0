"""Scattering lookup table container with the interface of lookup/lut.py:79-245 (`axes`, `axes_names`,
`axes_limits`, `axes_step`, `value_table`, `lookup_line`, `lookup_pts`, int16 packing).

The table itself is consumed on the GPU (uploaded once per class as float32); lookup_line / lookup_pts are kept
for API compatibility and host-side inspection: nearest node with np.round (half to even) on float32 axis
arithmetic, clamped (lut.py:240-241); multilinear interpolation through scipy.ndimage (lut.py:185-208).
"""
import numpy as np

INT_MIN, INT_MAX = -32768, 32767


class Error(Exception):
    pass


def float_to_uint16(x, use_log=False):
    """Pack a float array into int16 codes (lut.py:51-72): NaN / +inf -> -32768, the finite range mapped linearly
    onto -32767..32767, optionally after log(x + eps + min).  Truncating conversion, as astype does.  Returns
    {'data', 'use_log', 'log_offset', 'range'}; the argument is not modified."""
    x = np.array(x, dtype=np.float64, copy=True)
    log_offset = 0
    if use_log:
        log_offset = np.finfo(float).eps + np.nanmin(x)
        x = np.log(x + log_offset)
    x[x == np.inf] = np.nan
    lo, hi = np.nanmin(x), np.nanmax(x)
    ok = ~np.isnan(x)
    code = np.zeros(x.shape)
    code[ok] = (INT_MIN + 1) + (x[ok] - lo) / (hi - lo) * (INT_MAX - (INT_MIN + 1))
    code[~ok] = INT_MIN
    return {'data': code.astype('int16'), 'use_log': use_log, 'log_offset': log_offset, 'range': [lo, hi]}


def uint16_to_float(packed):
    """Inverse of float_to_uint16 up to the quantisation (lut.py:169-183)."""
    lo, hi = packed['range']
    c = np.asarray(packed['data']).astype('float64')
    x = np.zeros(c.shape)
    ok = c != INT_MIN
    x[ok] = lo + (c[ok] - (INT_MIN + 1)) / (INT_MAX - (INT_MIN + 1)) * (hi - lo)
    x[~ok] = np.nan
    if packed['use_log']:
        x = np.exp(x) - packed['log_offset']
    return x


class Lookup_table(object):
    def __init__(self):
        self.axes_names = {}
        self.axes_limits = []
        self.axes_step = []
        self.axes = []
        self.value_table = []
        self.int_transform = False
        self.log_offset = None
        self.use_log = False
        self.range_orig_data = None

    def add_axis(self, name, axis_values=None, is_regular=False):
        if name in self.axes_names:
            raise Error("Axis already exists with name: '%s'" % name)
        self.axes_names[name] = len(self.axes)
        v = np.asarray(axis_values).astype('float32')
        self.axes_limits.append([np.min(v), np.max(v)])
        self.axes_step.append(v[1] - v[0])
        self.axes.append(v)

    def set_axis_values(self, axis_name, axis_values, is_regular=False):
        if self.value_table is not None:
            raise Error("Cannot define axis once value table has been set.")
        i = self.axes_names[axis_name]
        v = np.asarray(axis_values).astype('float32')
        self.axes_limits[i] = [np.min(v), np.max(v)]
        self.axes[i] = v

    def get_axis_name(self, axis_i):
        for name, i in self.axes_names.items():
            if i == axis_i:
                return name
        return None

    def set_value_table(self, value_table, map_to_int=False):
        if map_to_int:
            m = float_to_uint16(value_table)
            self.int_transform, self.use_log = True, m['use_log']
            self.log_offset, self.range_orig_data = m['log_offset'], m['range']
            value_table = m['data']
        self.value_table = value_table if isinstance(value_table, np.ndarray) else np.array(value_table)

    def get_float_data(self, x):
        return uint16_to_float({'data': x, 'range': self.range_orig_data, 'use_log': self.use_log,
                                'log_offset': self.log_offset})

    def float_table(self):
        """The table as float32 whatever its storage (what the GPU takes)."""
        if self.int_transform:
            return self.get_float_data(self.value_table).astype(np.float32)
        return np.ascontiguousarray(self.value_table, dtype=np.float32)

    def lookup_pts(self, coords, mode='nearest', order=1):
        """Interpolated values at coords[n_axes, n_points] given in axis units (lut.py:185-208)."""
        from scipy import ndimage
        if not len(self.value_table):
            raise Error("No values set for lookup table")
        if coords.shape[0] != len(self.axes_names):
            raise Error("Invalid coordinates dimension")
        idx = [(c - lo) * (n - 1) / (hi - lo) for (lo, hi), c, n in zip(self.axes_limits, coords, self.value_table.shape)]
        out = ndimage.map_coordinates(self.value_table, idx, mode=mode, order=order)
        return self.get_float_data(out) if self.int_transform else out

    def node_index(self, name, x):
        ax = self.axes_names[name]
        c = np.round((np.asarray(x) - self.axes_limits[ax][0]) / self.axes_step[ax])
        return np.minimum(np.maximum(0, c), self.value_table.shape[ax] - 1).astype(int)

    def lookup_line(self, **kwargs):
        """Nearest-node slices, one per entry of the keyword arrays: lookup_line(e=..., t=...) ->
        [n, n_d, 12].  (Stored codes are returned as they are, like lut.py:232-245.)"""
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
