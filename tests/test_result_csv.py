"""The result table (LDPC_GallagerIntStreamSimu.java:186-291, LDPC_LoadIntStream.java:227-271): number
formatting as java.util.Formatter documents it, the derived rates with Java's division corner cases, and
the layout.  The reference ships no output file, so expected strings are worked out from its format strings."""
import numpy as np


def test_java_number_formatting(pkg):
    rc = pkg.result_csv
    assert rc.jfmt_f(0.75, 6) == "0.750000"
    assert rc.jfmt_f(0.125, 2) == "0.13"              # half up, where C's printf gives 0.12
    assert rc.jfmt_f(0.001, 3) == "0.001"
    assert rc.jfmt_f(2.5, 0) == "3"
    assert rc.jfmt_f(float("nan"), 4) == "NaN"
    assert rc.jfmt_f(float("inf"), 2) == "Infinity"
    assert rc.jfmt_e(0.0, 6) == "0.000000e+00"
    assert rc.jfmt_e(0.0123456789, 6) == "1.234568e-02"
    assert rc.jfmt_e(1234.5, 10) == "1.2345000000e+03"
    assert rc.jfmt_e(9.9999996e-4, 6) == "1.000000e-03"   # carry into the exponent
    assert rc.jfmt_e(1.5e-7, 0) == "2e-07"
    assert rc.jfmt_e(1e100, 2) == "1.00e+100"
    assert rc.jfmt_e(float("nan"), 6, 10) == "       NaN"   # "%10e": the width pads the word
    assert rc.jfmt_e(float("nan"), 6, 6) == "   NaN"
    assert rc.jfmt_e(float("-inf"), 6) == "-Infinity"


def _counters(pkg, L, frames, wrong, info_bits, chk_bits, info_fr, chk_fr, und, mis, it_ok, it_bad, ch, ch2, hist):
    c = np.zeros(pkg.capi.CNT_FIXED + L, np.uint64)
    c[:12] = [frames, wrong, info_bits, chk_bits, info_fr, chk_fr, und, mis, it_ok, it_bad, ch, ch2]
    c[pkg.capi.CNT_FIXED:] = hist
    return c


def test_leg_statistics_and_corner_cases(pkg):
    rc = pkg.result_csv
    L = 5
    s = rc.leg_statistics(_counters(pkg, L, 100, 4, 30, 6, 3, 2, 1, 12, 192, 20, 500, 2700, [50, 40, 6, 0, 4]), 4, 12)
    assert s["FER"] == 0.04 and s["IFER"] == 0.03 and s["CFER"] == 0.02 and s["avg_error_check_bits"] == 3.0
    assert s["IBER"] == 30 / 800 and s["avg_iters_correct"] == 2.0 and s["avg_iters_wrong"] == 5.0
    assert s["mis_rate"] == 0.4 and s["undetected"] == 1 and s["histogram"] == [50, 40, 6, 0, 4]
    assert abs(s["channel_ber_mean"] - 500 / 800) < 1e-15
    assert abs(s["channel_ber_var"] - (2700 / 64 / 100 - (500 / 800) ** 2)) < 1e-12
    # a clean leg: 0/0 -> NaN (:209), no failed frame -> NaN (:193-194), misRate 0.0 (:247)
    z = rc.leg_statistics(_counters(pkg, L, 10, 0, 0, 0, 0, 0, 0, 0, 10, 0, 0, 0, [10, 0, 0, 0, 0]), 4, 12)
    assert z["FER"] == 0.0 and z["avg_error_check_bits"] != z["avg_error_check_bits"]
    assert z["avg_iters_wrong"] != z["avg_iters_wrong"] and z["mis_rate"] == 0.0 and z["avg_iters_correct"] == 1.0


def test_multi_matrix_layout(pkg, tmp_path):
    rc = pkg.result_csv
    L, ne, ncm = 4, 3, 2
    rng = np.random.default_rng(0)
    cnt = [[_counters(pkg, L, 100, 5 + i, 40 + j, 7, 4, 2, 1, 9, 150, 20, 300 + i, 1200, rng.integers(0, 30, L))
            for i in range(ne)] for j in range(ncm)]
    cnt[1][0] = _counters(pkg, L, 100, 0, 0, 0, 0, 0, 0, 0, 100, 0, 0, 0, [100, 0, 0, 0])
    e = [0.001, 0.02, 0.05]
    text = rc.write_result_csv(tmp_path / "x-result.csv", 16, 8, 2, L, 100, e, cnt, total_minutes=[2.0, 4.0],
                               decode_minutes=[[0.5, 0.25, 0.25], [1.0, 1.0, 1.0]], encoding="cp932")
    assert (tmp_path / "x-result.csv").read_bytes() == text.encode("cp932")       # the Java tool's charset
    lines = text.split("\n")
    assert lines[0] == (",".join(rc.HEAD_LABELS) + ",,,,,,") * 2
    assert lines[1] == "16,8,2,0.750000,4,100,2.00,1.00,50.00,0,,,,,,16,8,2,0.750000,4,100,4.00,3.00,75.00,1,,,,,,"
    assert lines[2] == "" and lines[3] == (",".join(rc.ROW_LABELS) + ",,") * 2
    first = lines[4].split(",")
    assert len(first) == 2 * 15 + 1 and first[0] == "0.001" and first[3] == "5.000000e-02"
    assert first[12] == "(9/40)" and first[13] == "0.50" and first[14] == ""
    clean = first[15:29]                                                           # matrix 1, first error rate
    assert clean[3] == "0.000000e+00" and clean[6] == "       NaN" and clean[9] == "1.0000" and clean[10] == "NaN"
    assert clean[11] == "0.000000e+00" and clean[12] == "(0/0)"
    assert lines[4 + ne] == "" and lines[5 + ne] == rc.HIST_TITLE
    assert lines[6 + ne] == (rc.HIST_CORNER + ",0.00,0.02,0.05,,,") * 2            # "%.2f" of the error rates
    rows, hist = rc.parse_result_csv(text, ncm, ne, L)
    for j in range(ncm):
        for i in range(ne):
            assert (hist[j, i] == cnt[j][i][pkg.capi.CNT_FIXED:].astype(np.int64)).all()
            assert abs(float(rows[j][i][3]) - int(cnt[j][i][1]) / 100) < 1e-9
    assert lines[7 + ne + L - 1].startswith("%d," % (L - 1)) and lines[7 + ne + L] == ""


def test_single_matrix_layout(pkg):
    rc = pkg.result_csv
    L = 3
    cnt = [_counters(pkg, L, 50, 5, 40, 7, 4, 2, 1, 9, 90, 15, 300, 2000, [30, 15, 5])]
    text = rc.format_result_csv_single(12, 6, 2, L, 50, [0.03], cnt, 4, total_minutes=1.0, decode_minutes=[0.5])
    lines = text.split("\n")
    assert lines[0] == ",".join(rc.HEAD_LABELS[:9])
    assert lines[1] == "12,6,2,0.666667,3,50,1.00,0.50,50.00" and lines[2] == ""
    assert lines[3] == ",".join(rc.ROW_LABELS[:12])       # the format string there has 12 conversions for 14 labels
    cells = lines[4].split(",")
    assert cells[:2] == ["0.030", "7.50000000e-01"] and cells[3] == "1.00000000e-01" and cells[13] == "0.500000"
    assert lines[6] == rc.HIST_TITLE and lines[7] == rc.HIST_CORNER + ",0.03,"
    assert lines[8:11] == ["0,30,", "1,15,", "2,5,"]
