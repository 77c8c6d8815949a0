"""cosmo_pol_amd: MI355X (gfx950) implementation of the cosmo_pol per-gate forward-operator hot path.

Public names follow the reference package (cosmo_pol): RadarOperator, interpolation.get_profiles_GH,
scattering_sz.get_radar_observables, doppler_sz.get_doppler_velocity, lut.Lookup_table, cfg, constants,
beam.Beam, plus interpolation_c / doppler_c (the SWIG modules' replacements).  Everything numerical runs in
libcosmopol_hip.so (cosmo_pol_amd/csrc, C ABI in include/cosmopol.h); importing the package does not load
the library, using any operator does -- and fails loudly if it is missing."""
__all__ = ['RadarOperator']


def __getattr__(name):
    if name == 'RadarOperator':
        from .radar_operator import RadarOperator
        return RadarOperator
    raise AttributeError(name)
