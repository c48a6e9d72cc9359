#!/usr/bin/env python3
"""What 4-byte message storage alone, and float arithmetic on top of it, do to the decodes of an irregular code
(CPU only; the double oracle against itself with parts switched to float, oracle/ldpc_oracle.c `stored`).

NR-BG1-shaped Z = 16 code (n = 1088, check degrees 3..19, variable degrees 1..30), the 20,000 + 12,000 frames of
tests/test_gpu_parity.py::test_hard_decision_agreement_at_scale_irregular, early termination, 50 iterations.
For every mode: frames whose (word, iterations, syndrome weight) differ from the all-double decode
(LogDecoder.java:64-120 restated), split into frames the double decode brings to a zero syndrome and frames it
does not.

    python tools/fp32_storage_study.py > profiles/r03/fp32_storage_study.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ldpc_simulator_amd as L
from oracle import oracle_capi as oc

MODES = ((1, "messages stored as float (nats), arithmetic double"),
         (2, "messages stored as float (bits), arithmetic double"),
         (6, "+ variable-node sums in float (prefix + suffix)"),
         (10, "+ check-node rule on complements in float, variable-node sums double"),
         (14, "+ both: the model of the float implementation"))


def main():
    _, (m, n, rows, cols) = L.codes.nr_bg1_shaped(z=16, seed=4)
    g = oc.Graph(m, n, rows, cols)
    total = {mode: [0, 0] for mode, _ in MODES}
    union = 0
    for ebn0, F in ((2.0, 20000), (1.0, 12000)):
        llr = oc.gen_awgn_llr(m, n, ebn0, seed=33, first_frame=0, frames=F)
        want = g.decode_llr_batch(llr, 50, True)
        conv = want[2] == 0
        print("Eb/N0 %.1f dB, %d frames, the double decode leaves %d undecoded" % (ebn0, F, int((~conv).sum())))
        any_diff = np.zeros(F, bool)
        for mode, text in MODES:
            got = g.decode_llr_batch(llr, 50, True, storage=mode)
            same = (got[0] == want[0]).all(axis=1) & (got[1] == want[1]) & (got[2] == want[2])
            d = np.nonzero(~same)[0]
            any_diff |= ~same
            total[mode][0] += int(conv[d].sum())
            total[mode][1] += int((~conv[d]).sum())
            print("  mode %2d (%s): %d frames differ, %d of them among the decoded; (frame, bits, weight / weight of the double decode): %s"
                  % (mode, text, d.size, int(conv[d].sum()),
                     [(int(f), int((got[0][f] != want[0][f]).sum()), int(got[2][f]), int(want[2][f])) for f in d]))
            sys.stdout.flush()
        union += int(any_diff.sum())
    print("all 32,000 frames: mode -> (decoded frames that differ, undecoded frames that differ):",
          {mode: tuple(v) for mode, v in total.items()}, "; frames that differ in at least one mode:", union)


if __name__ == "__main__":
    main()
