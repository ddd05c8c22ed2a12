"""lgmi — Python host of the MI355X-native site-pair mutual-information engine.

Mirrors the reference interface of this path (gxiaolab/L-GIREMI,
src/giremi/mutual_information.py) on top of liblgmi.so (C ABI, include/lgmi.h).
"""
from .mutual_information import (mean_mismatch_pair_mutual_info, mismatch_pair_mutual_info,  # noqa: F401
                                 region_pair_mi, regions_pair_mi)
from .engine import Engine, MIResult, default_engine, default_synth_spec, make_params, plan_shard  # noqa: F401
from .pack import PackedBatch, pack_blocks  # noqa: F401
from .stat import ecdf, mean_mi_to_mip  # noqa: F401
from . import dist, synth  # noqa: F401
from .splice import site_splice_mi, site_splice_pairs  # noqa: F401
from .region import (get_region_mismatches_with_filters, region_mismatch_analysis,  # noqa: F401
                     regions_mismatch_analysis)

__version__ = '0.1.0'
