"""Process-wide engine cache: the reference keeps configuration, model state and lookup tables
in module globals (utilities/cfg.py:9, radar_operator.py:71-84); the GPU context plays that role
here.  One engine per (device); model / tables are re-uploaded only when the caller hands in a
different object or asks for it (force=True; in-place edits of the arrays are not detected otherwise)."""
import copy

import numpy as np

from . import cfg, engine

_engines = {}


class _State(object):
    def __init__(self):
        self.eng = None
        self.conf_snapshot = None
        self.model_key = None
        self.lut_key = None
        self.quad_key = None


def get_engine(device=0, doppler=None, sensitivity_cut=False):
    st = _engines.setdefault(device, _State())
    conf = cfg.CONFIG if cfg.CONFIG else cfg.DEFAULTS
    snap = (copy.deepcopy(conf), doppler, sensitivity_cut)
    if st.eng is None:
        st.eng = engine.Engine(conf, device=device)
        st.conf_snapshot = None
    if st.conf_snapshot != snap:
        st.eng.configure(conf, doppler=doppler, sensitivity_cut=sensitivity_cut)
        st.conf_snapshot = snap
        st.quad_key = None
        if st.lut_key is not None and st.lut_key[1] != conf['microphysics']['scheme']:
            st.lut_key = None
    return st


def ensure_luts(st, lut_sz, force=False):
    # the uploaded object itself is kept and compared with `is`: an id() can be reused by a later object once the
    # earlier one has been freed
    scheme = st.eng.conf['microphysics']['scheme']
    if force or st.lut_key is None or st.lut_key[0] is not lut_sz or st.lut_key[1] != scheme:
        st.eng.upload_luts(lut_sz)
        st.lut_key = (lut_sz, scheme)


def ensure_model(st, dic_variables, force=False):
    if force or st.model_key is not dic_variables:
        first = next(iter(dic_variables.values()))
        fields = {k: v.data for k, v in dic_variables.items()}
        st.eng.upload_model(fields, first.attributes['z-levels'], first.attributes['proj_info'],
                            first.attributes['resolution'])
        st.model_key = dic_variables


def ensure_quadrature(st):
    integ = st.eng.conf['integration']
    key = (repr(sorted((k, np.asarray(v).tolist()) for k, v in integ.items())), st.eng.conf['radar']['3dB_beamwidth'])
    if st.quad_key != key:
        st.quad = st.eng.set_gauss_hermite()      # every integration scheme ends in the same sub-beam list
        st.quad_key = key
    return st.quad


def reset():
    for st in _engines.values():
        if st.eng is not None:
            st.eng.ctx.close()
    _engines.clear()
