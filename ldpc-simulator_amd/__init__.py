"""MI355X-native flooding belief-propagation LDPC decoder (hot path of LDPC-Simulator).

capi           ctypes binding of the C ABI (include/ldpc_amd.h -> libldpc_amd.so)
log_decoder    LogDecoder / LogDecoderBefore / MinSumDecoder with the reference's signature
codes          seeded parity-check matrix builders
encoder        generator matrix, systematic form, column-permuted H (the reference's encoder chain)
check_matrix_io  the reference's H text format
sim            frame sharding over GPUs and the counter all-reduce
result_csv     counters -> the reference's result table (same columns, labels and printf conversions)
drivers        the reference's simulation mains as functions (build/load code, run the legs, write the files)
"""
from . import capi, check_matrix_io, codes, drivers, encoder, result_csv, sim  # noqa: F401
from .capi import Decoder, LdpcError  # noqa: F401
from .log_decoder import DecodeResult, LogDecoder, LogDecoderBefore, MinSumDecoder  # noqa: F401
