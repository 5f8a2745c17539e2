"""HBV 1.0 on the MI355X-native time-stepper.

Drop-in for the reference plug-in `hydrodl2.load_model('hbv')`
(src/hydrodl2/models/hbv/hbv.py:8-596): same constructor, attributes, state API
and flux dictionary; the per-day loop, the parameter prep and the routing run in
the HIP library behind include/hbvx.h.
"""
from hydrodl2_amd import _abi
from hydrodl2_amd.core.hbv_module import HbvModule


class Hbv(HbvModule):
    """HBV 1.0: 12 physical parameters (+ parBETAET iff listed dynamic) x nmul, 2 routing."""

    _model_id = _abi.MODEL_HBV10
    _display_name = 'HBV 1.0'
    _extra_bounds = {}
    _has_capillary = False
