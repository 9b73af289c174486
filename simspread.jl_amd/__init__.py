"""simspread.jl_amd -- MI355X-native SimSpread resource-spreading engine.

The hot path ``featurize -> construct -> spread -> predict -> clean!`` of cvigilv/SimSpread.jl
(src/core.jl) behind the reference's own function names, computed by hand-written gfx950
kernels in ``libsimspread_hip.so`` (C ABI: include/simspread_hip.h).  No CPU fallback.
"""
from . import _lib
from ._lib import SimSpreadError, init, path_last, timing_hold, timing_last, use_torch_stream
from .core import (NamedMatrix, Network, clean, clean_, construct, cutoff, featurize, k, names, precisionatL, predict,
                   read_namedmatrix, recallatL, save, save_loo, split, spread, writedlm)
from .dist import comm_destroy, comm_init, comm_unique_id, gather_scores, gather_topl, lib_gather_scores, shard_range
from .engine import DeviceGraph, DeviceSpMat, jaccard_similarity, rank_metrics, topl
from .metrics import (AuPRC, AuROC, BEDROC, ROCNums, accuracy, balancedaccuracy, f1score, maxperformance, mcc,
                      meanperformance, meanstdperformance, precision, recall, roc, validity_ratio)

__all__ = ["NamedMatrix", "Network", "DeviceGraph", "DeviceSpMat", "SimSpreadError", "init", "timing_last", "timing_hold", "path_last", "use_torch_stream",
           "shard_range", "gather_scores", "gather_topl", "comm_unique_id", "comm_init", "comm_destroy", "lib_gather_scores", "k", "cutoff", "featurize", "construct", "spread", "predict", "clean", "clean_", "names", "split", "save", "save_loo", "topl", "recallatL", "precisionatL",
           "read_namedmatrix", "writedlm", "rank_metrics", "jaccard_similarity", "AuROC", "AuPRC", "BEDROC", "validity_ratio", "roc", "ROCNums",
           "f1score", "mcc", "accuracy", "balancedaccuracy", "recall", "precision", "maxperformance", "meanperformance",
           "meanstdperformance"]
