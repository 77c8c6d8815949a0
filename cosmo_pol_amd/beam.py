"""Beam record with the attribute names of utilities/beam.py:8-18."""


class Beam(object):
    def __init__(self, dic_values, mask, lats_profile, lons_profile, dist_ground_profile, heights_profile,
                 elev_profile=[], GH_pt=[], GH_weight=1):
        self.values = dic_values
        self.mask = mask
        self.lats_profile = lats_profile
        self.lons_profile = lons_profile
        self.dist_profile = dist_ground_profile
        self.heights_profile = heights_profile
        self.elev_profile = elev_profile
        self.GH_pt = GH_pt
        self.GH_weight = GH_weight
