"""The result table of the reference's drivers, from counter vectors.

LDPC_GallagerIntStreamSimu.java:186-212 turns the per-leg counters into rates and :231-291 prints
them: one block of columns per parity-check matrix side by side, one row per channel error rate,
then the histogram of iteration counts.  This module does both steps for the counter vector of
ldpc_amd_simulate (capi.CNT_*), keeping the reference's column order, labels, printf conversions and
its floating-point corner cases (x/0 -> NaN or Infinity, printed as java.util.Formatter prints them).

The reference writes Windows-31j; the default here is UTF-8 and `encoding="cp932"` gives the same
bytes as the Java tool.  No reference output file exists to compare with, so the layout is checked
against the source of the writer, not against a golden file.
"""
import math
from decimal import ROUND_HALF_UP, Decimal

import numpy as np

from . import capi

HEAD_LABELS = ("符号長", "行重み", "列重み", "符号化率", "最大反復回数", "フレーム数", "全体時間(m)", "合計復号時間(m)",
               "復号割合(%)", "行列番号")                                     # :233
ROW_LABELS = ("通信路誤り率", "実際の通信路誤り率の平均", "実際の通信路誤り率の分散", "FER", "IFER", "CFER",
              "平均誤り検査ビット数", "s=0だが誤訂正", "IBER", "成功時の平均繰り返し回数", "失敗時の平均繰り返し回数",
              "平均誤訂正ビット率", "誤訂正ビット/残留ビット", "各誤り率の復号時間(m)")  # :240-241
HIST_TITLE = "以下は反復回数の度数分布"                                       # :269
HIST_CORNER = "回数\\通信路誤り率"                                            # :271


# ---- java.util.Formatter for doubles -----------------------------------------------------------
# %f and %e round the shortest decimal string that identifies the double, half up; NaN and the
# infinities print as words and, like every conversion, are padded to the field width.
def _special(x):
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "Infinity" if x > 0 else "-Infinity"
    return None


def jfmt_f(x, prec=6, width=0):
    x = float(x)
    s = _special(x)
    if s is None:
        s = format(Decimal(repr(x)).quantize(Decimal(1).scaleb(-prec), rounding=ROUND_HALF_UP), "f")
    return s.rjust(width)


def jfmt_e(x, prec=6, width=0):
    x = float(x)
    s = _special(x)
    if s is None:
        d = Decimal(repr(x))
        if d.is_zero():
            mant, exp = d.quantize(Decimal(1).scaleb(-prec)), 0
        else:
            exp = d.adjusted()
            mant = d.scaleb(-exp).quantize(Decimal(1).scaleb(-prec), rounding=ROUND_HALF_UP)
            if abs(mant) >= 10:
                mant = (mant / 10).quantize(Decimal(1).scaleb(-prec), rounding=ROUND_HALF_UP)
                exp += 1
        s = "%se%s%02d" % (format(mant, "f"), "-" if exp < 0 else "+", abs(exp))
    return s.rjust(width)


def _jdiv(a, b):
    """Java's double division: finite/0 is an infinity, 0/0 is NaN."""
    a, b = float(a), float(b)
    if b == 0.0:
        return float("nan") if a == 0.0 or math.isnan(a) else math.copysign(float("inf"), a)
    return a / b


# ---- counters -> the numbers of one table cell group (:186-212, :220-228, :247) ---------------------
def leg_statistics(counters, m, n):
    c = [int(v) for v in np.asarray(counters).ravel()]
    frames = c[capi.CNT_FRAMES]
    k = n - m                                                          # g.length == gLength
    wrong = c[capi.CNT_FRAME_ERRORS]
    ok = frames - wrong
    info_bits, mis = c[capi.CNT_INFO_BIT_ERRORS], c[capi.CNT_MISCORRECTED]
    # per-frame channel error rate x_f = errors_f / k: mean = sum x_f / F, variance = sum (x_f - mean)^2 / F
    mean = _jdiv(_jdiv(c[capi.CNT_CHANNEL_ERRORS], k), frames)
    # The reference sums (x_f - mean)^2 over the frames (:221-228); from the two integer sums that is
    # (F * sum e_f^2 - (sum e_f)^2) / (k^2 F^2), formed in exact integer arithmetic and rounded once, instead of
    # the cancelling difference E[x^2] - mean^2 of two rounded doubles.
    S, SQ = c[capi.CNT_CHANNEL_ERRORS], c[capi.CNT_CHANNEL_ERRORS_SQ]
    var = (frames * SQ - S * S) / (k * k * frames * frames) if frames and k else float("nan")
    return {
        "channel_ber_mean": mean,
        "channel_ber_var": max(var, 0.0) if var == var else var,
        "FER": _jdiv(wrong, frames),
        "IFER": _jdiv(c[capi.CNT_INFO_FRAME_ERRORS], frames),
        "CFER": _jdiv(c[capi.CNT_CHECK_FRAME_ERRORS], frames),
        "avg_error_check_bits": _jdiv(c[capi.CNT_CHECK_BIT_ERRORS], c[capi.CNT_CHECK_FRAME_ERRORS]),
        "undetected": c[capi.CNT_UNDETECTED],
        "IBER": _jdiv(info_bits, k * frames),
        "avg_iters_correct": float("nan") if ok == 0 else c[capi.CNT_ITERS_CORRECT] / ok,       # :191-192
        "avg_iters_wrong": float("nan") if wrong == 0 else c[capi.CNT_ITERS_WRONG] / wrong,     # :193-194
        "mis_rate": 0.0 if info_bits == 0 else mis / info_bits,                                # :247
        "miscorrected_bits": mis,
        "residual_info_bits": info_bits,
        "histogram": c[capi.CNT_FIXED:],
    }


def format_result_csv(n, wr, wc, max_l, num_frames, e_values, counters, m=None, total_minutes=None,
                      decode_minutes=None):
    """counters[matrix][e] -> the text of `<name>-result.csv`.

    counters: nested sequence [numCM][len(e_values)] of uint64[CNT_FIXED + max_l] vectors.
    total_minutes[matrix], decode_minutes[matrix][e]: wall times in minutes (zeros when omitted).
    m defaults to the Gallager construction's n*wc/wr rows."""
    num_cm, ne = len(counters), len(e_values)
    m = (n * wc) // wr if m is None else m
    total_minutes = [0.0] * num_cm if total_minutes is None else total_minutes
    decode_minutes = [[0.0] * ne for _ in range(num_cm)] if decode_minutes is None else decode_minutes
    out = []
    out.append("".join(",".join(HEAD_LABELS) + ",,,,,," for _ in range(num_cm)) + "\n")       # :233-234
    for i in range(num_cm):                                                                     # :236-238
        dec = math.fsum(decode_minutes[i])
        share = _jdiv(dec, total_minutes[i]) * 100
        out.append("%d,%d,%d,%s,%d,%d,%s,%s,%s,%d,,,,,," % (
            n, wr, wc, jfmt_f(1 - wc / wr, 6), max_l, num_frames, jfmt_f(total_minutes[i], 2), jfmt_f(dec, 2),
            jfmt_f(share, 2), i))
    out.append("\n\n")
    out.append("".join(",".join(ROW_LABELS) + ",," for _ in range(num_cm)) + "\n")             # :240-242
    stats = [[leg_statistics(counters[j][i], m, n) for i in range(ne)] for j in range(num_cm)]
    for i in range(ne):                                                                         # :244-267
        for j in range(num_cm):
            s = stats[j][i]
            out.append(",".join((
                jfmt_f(e_values[i], 3), jfmt_e(s["channel_ber_mean"], 6), jfmt_e(s["channel_ber_var"], 10),
                jfmt_e(s["FER"], 6), jfmt_e(s["IFER"], 6), jfmt_e(s["CFER"], 6, 6),
                jfmt_e(s["avg_error_check_bits"], 6, 10), "%d" % s["undetected"], jfmt_e(s["IBER"], 6),
                jfmt_f(s["avg_iters_correct"], 4), jfmt_f(s["avg_iters_wrong"], 4), jfmt_e(s["mis_rate"], 6),
                "(%d/%d)" % (s["miscorrected_bits"], s["residual_info_bits"]), jfmt_f(decode_minutes[j][i], 2))) + ",,")
        out.append("\n")
    out.append("\n" + HIST_TITLE + "\n")                                                        # :269
    for _ in range(num_cm):                                                                     # :270-277
        out.append(HIST_CORNER + "," + "".join(jfmt_f(e, 2) + "," for e in e_values) + ",,")
    out.append("\n")
    for it in range(max_l):                                                                     # :279-288
        for j in range(num_cm):
            out.append("%d," % it + "".join("%d," % stats[j][k]["histogram"][it] for k in range(ne)) + ",,")
        out.append("\n")
    return "".join(out)


def format_result_csv_single(n, wr, wc, max_l, num_frames, e_values, counters, m, total_minutes=0.0,
                             decode_minutes=None):
    """The one-matrix table of the drivers that load H from a file (LDPC_LoadIntStream.java:227-271):
    no matrix index, %.8e cells, and a label row that stops after 12 of the 14 labels because its format
    string has 12 conversions.  counters[e] are the legs of the one matrix."""
    ne = len(e_values)
    decode_minutes = [0.0] * ne if decode_minutes is None else decode_minutes
    dec = math.fsum(decode_minutes)
    out = [",".join(HEAD_LABELS[:9]) + "\n"]                                                   # :229-230
    out.append("%d,%d,%d,%s,%d,%d,%s,%s,%s\n\n" % (n, wr, wc, jfmt_f(1 - wc / wr, 6), max_l, num_frames,
                                                   jfmt_f(total_minutes, 2), jfmt_f(dec, 2),
                                                   jfmt_f(_jdiv(dec, total_minutes) * 100, 2)))  # :232-233
    out.append(",".join(ROW_LABELS[:12]) + "\n")                                               # :235-239
    stats = [leg_statistics(counters[i], m, n) for i in range(ne)]
    for i, s in enumerate(stats):                                                              # :241-260
        out.append(",".join((
            jfmt_f(e_values[i], 3), jfmt_e(s["channel_ber_mean"], 8), jfmt_e(s["channel_ber_var"], 10),
            jfmt_e(s["FER"], 8), jfmt_e(s["IFER"], 8), jfmt_e(s["CFER"], 8), jfmt_e(s["avg_error_check_bits"], 8),
            "%d" % s["undetected"], jfmt_e(s["IBER"], 8), jfmt_f(s["avg_iters_correct"], 4),
            jfmt_f(s["avg_iters_wrong"], 4), jfmt_e(s["mis_rate"], 8),
            "(%d/%d)" % (s["miscorrected_bits"], s["residual_info_bits"]), jfmt_f(decode_minutes[i], 6))) + ",,\n")
    out.append("\n" + HIST_TITLE + "\n" + HIST_CORNER + "," + "".join(jfmt_f(e, 2) + "," for e in e_values) + "\n")
    for it in range(max_l):                                                                    # :267-271
        out.append("%d," % it + "".join("%d," % stats[k]["histogram"][it] for k in range(ne)) + "\n")
    return "".join(out)


def write_result_csv(path, *args, encoding="utf-8", single=False, **kwargs):
    text = (format_result_csv_single if single else format_result_csv)(*args, **kwargs)
    with open(path, "w", encoding=encoding, newline="") as f:
        f.write(text)
    return text


def parse_result_csv(text, num_cm, num_e, max_l):
    """Back from the table to numbers (strings for the '(a/b)' cells): a list per matrix of rows, and the
    histogram [matrix][e][iteration-1].  For tests and for comparing two tools' outputs."""
    lines = text.split("\n")
    rows = [[None] * num_e for _ in range(num_cm)]
    stride = len(ROW_LABELS) + 1            # 14 cells and the empty one between the two closing commas
    for i in range(num_e):
        cells = lines[4 + i].split(",")
        for j in range(num_cm):
            rows[j][i] = cells[j * stride: j * stride + len(ROW_LABELS)]
    h0 = 4 + num_e + 3
    hist = np.zeros((num_cm, num_e, max_l), np.int64)
    hstride = 1 + num_e + 2
    for it in range(max_l):
        cells = lines[h0 + it].split(",")
        for j in range(num_cm):
            assert int(cells[j * hstride]) == it
            hist[j, :, it] = [int(v) for v in cells[j * hstride + 1: j * hstride + 1 + num_e]]
    return rows, hist
