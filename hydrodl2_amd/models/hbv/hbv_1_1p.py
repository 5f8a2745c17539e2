"""HBV 1.1p (capillary rise + always-on parBETAET) on the MI355X-native time-stepper.

Drop-in for `hydrodl2.load_model('hbv_1_1p')`
(src/hydrodl2/models/hbv/hbv_1_1p.py:8-608).
"""
from hydrodl2_amd import _abi
from hydrodl2_amd.core.hbv_module import HbvModule


class Hbv_1_1p(HbvModule):
    """HBV 1.1p: 14 physical parameters x nmul, 2 routing (hbv_1_1p.py:87-106)."""

    _model_id = _abi.MODEL_HBV11P
    _display_name = 'HBV 1.1p'
    _extra_bounds = {'parBETAET': [0.3, 5], 'parC': [0, 1]}
    _has_capillary = True
