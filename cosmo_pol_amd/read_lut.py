"""Lookup-table files: get_lookup_tables(scheme, freq) with the meaning of lookup/read_lut.py:10-20, on a flat
binary format instead of the reference's gzip-compressed serialized Python-2 objects.

File `all_luts_SZ_f_<freq>_<scheme>mom.clut` (one per frequency and microphysics scheme):

    bytes 0..7     magic  b'CPOLLUT1'
    bytes 8..15    little-endian uint64: length n of the JSON header
    bytes 16..16+n JSON: {"frequency", "scheme", "classes": {"R": {"axes": {"e": [...], "t": [...], "d": [...],
                   "sz": [...]}, "order": ["e","t","d","sz"], "shape", "dtype": "<f4" | "<i2", "offset", "nbytes",
                   "packing": null | {"range", "use_log", "log_offset"}}, ...}}
    then, 4096-byte aligned, each class's table [E, T, D, 12] in C order

so a table is one np.memmap away from the host pointer cpol_upload_lut takes, and nothing in the file is
executable.  int16 tables use the reference's packing (lookup/lut.py:51-72,169-183).

Tables that exist only in the reference's own format are exported from an installation of the reference
(arrays written with numpy.save, see from_arrays_dir) and read here as plain arrays; this package never
deserializes Python objects.
"""
import json
import os
import struct

import numpy as np

from .lut import Lookup_table, float_to_uint16

MAGIC = b'CPOLLUT1'
ALIGN = 4096
FREQUENCIES = [2.7, 4.15, 5.6, 7.7, 9.8, 11.7, 13.6, 24.6, 35.6]       # lookup/compute_lut_sz.py:25
FOLDER_LUT = os.path.join(os.path.dirname(os.path.realpath(__file__)), 'final_lut')


def lut_filename(scheme, freq, folder=None):
    return os.path.join(folder or FOLDER_LUT, 'all_luts_SZ_f_%s_%smom.clut' % (str(freq).replace('.', '_'), scheme))


def save_lookup_tables(path, lut_sz, frequency=None, scheme=None, pack_int16=False):
    """Write {class letter: Lookup_table} to one .clut file."""
    classes, blobs = {}, []
    offset = 0
    for h, lut in lut_sz.items():
        names = [lut.get_axis_name(i) for i in range(len(lut.axes))]
        tab = lut.value_table
        packing = None
        if lut.int_transform:
            packing = {'range': [float(v) for v in lut.range_orig_data], 'use_log': bool(lut.use_log),
                       'log_offset': float(lut.log_offset)}
        elif pack_int16:
            m = float_to_uint16(tab)
            tab = m['data']
            packing = {'range': [float(v) for v in m['range']], 'use_log': bool(m['use_log']),
                       'log_offset': float(m['log_offset'])}
        tab = np.ascontiguousarray(tab, dtype='<i2' if packing else '<f4')
        classes[h] = {'axes': {n: [float(v) for v in lut.axes[lut.axes_names[n]]] for n in names}, 'order': names,
                      'shape': list(tab.shape), 'dtype': tab.dtype.str, 'offset': offset, 'nbytes': tab.nbytes,
                      'packing': packing}
        blobs.append(tab)
        offset += -(-tab.nbytes // ALIGN) * ALIGN
    header = json.dumps({'frequency': frequency, 'scheme': scheme, 'classes': classes}).encode()
    data_start = -(-(16 + len(header)) // ALIGN) * ALIGN
    with open(path, 'wb') as fh:
        fh.write(MAGIC + struct.pack('<Q', len(header)) + header)
        for h, tab in zip(classes, blobs):
            fh.seek(data_start + classes[h]['offset'])
            fh.write(tab.tobytes())
        fh.truncate(data_start + offset)


def load_lookup_tables(path, mmap=True):
    """{class letter: Lookup_table} from a .clut file; value tables are read-only memory maps by default."""
    with open(path, 'rb') as fh:
        if fh.read(8) != MAGIC:
            raise ValueError('%s is not a lookup-table file of this package' % path)
        (n,) = struct.unpack('<Q', fh.read(8))
        header = json.loads(fh.read(n).decode())
    data_start = -(-(16 + n) // ALIGN) * ALIGN
    out = {}
    for h, c in header['classes'].items():
        lut = Lookup_table()
        for name in c['order']:
            lut.add_axis(name, np.asarray(c['axes'][name]))
        if mmap:
            tab = np.memmap(path, dtype=np.dtype(c['dtype']), mode='r', offset=data_start + c['offset'],
                            shape=tuple(c['shape']))
        else:
            with open(path, 'rb') as fh:
                fh.seek(data_start + c['offset'])
                tab = np.frombuffer(fh.read(c['nbytes']), dtype=np.dtype(c['dtype'])).reshape(c['shape'])
        lut.value_table = tab
        if c['packing']:
            lut.int_transform, lut.use_log = True, c['packing']['use_log']
            lut.log_offset, lut.range_orig_data = c['packing']['log_offset'], c['packing']['range']
        out[h] = lut
    return out


def get_lookup_tables(scheme, freq, load_rcs=False, folder=None):
    """Tables of one frequency and scheme (1 / 2 or '1mom' / '2mom'); when the frequency has no file, the two
    closest tabulated frequencies are interpolated linearly once and the result is stored
    (lookup/read_lut.py:10-56 -- the reference's interpolation refers to files and methods that do not exist, so
    this is its evident intent rather than its letter)."""
    scheme = str(scheme).replace('mom', '')
    path = lut_filename(scheme, freq, folder)
    if not os.path.exists(path):
        interpolate_lookup_tables(scheme, freq, folder)
    return load_lookup_tables(path)


def interpolate_lookup_tables(scheme, freq, folder=None):
    i = int(np.searchsorted(FREQUENCIES, freq))
    if i == 0 or i == len(FREQUENCIES):
        raise ValueError('frequency %s GHz is outside the tabulated range %s' % (freq, FREQUENCIES))
    f1, f2 = FREQUENCIES[i - 1], FREQUENCIES[i]
    p1, p2 = lut_filename(scheme, f1, folder), lut_filename(scheme, f2, folder)
    for p in (p1, p2):
        if not os.path.exists(p):
            raise IOError('no lookup tables for %s GHz and none at the neighbouring frequency: %s is missing' % (freq, p))
    l1, l2 = load_lookup_tables(p1), load_lookup_tables(p2)
    out = {}
    for h in l1:
        t1, t2 = l1[h].float_table().astype(np.float64), l2[h].float_table().astype(np.float64)
        lut = Lookup_table()
        for k in range(len(l1[h].axes)):
            lut.add_axis(l1[h].get_axis_name(k), l1[h].axes[k])
        lut.set_value_table((t1 + (t2 - t1) / (f2 - f1) * (freq - f1)).astype(np.float32))
        out[h] = lut
    save_lookup_tables(lut_filename(scheme, freq, folder), out, frequency=freq, scheme=scheme)


def from_arrays_dir(folder, classes=('R', 'S', 'G', 'H')):
    """Tables exported as plain arrays, one set per class: <h>_e.npy, <h>_t.npy, <h>_d.npy, <h>_table.npy
    ([E, T, D, 12]); classes without files are skipped."""
    out = {}
    for h in classes:
        p = lambda s: os.path.join(folder, '%s_%s.npy' % (h, s))
        if os.path.exists(p('table')):
            out[h] = Lookup_table.from_arrays(np.load(p('e')), np.load(p('t')), np.load(p('d')), np.load(p('table')))
    return out
