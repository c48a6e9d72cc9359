"""The reference's mains run end to end on the GPU (LDPC_GallagerIntStreamSimu.java:10-293,
LDPC_LoadIntStream.java:10-276): files written, table consistent with the counters, counters consistent
with the CPU checker run on the same seeded legs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("decoder", ["LogDecoder", "LogDecoderBefore", "MinSumDecoder"])
def test_gallager_int_stream_simu(pkg, oracle, tmp_path, decoder):
    n, wr, wc, L, F, ncm = 256, 8, 2, 20, 600, 2
    e_values = (0.01, 0.04)
    cnt = pkg.drivers.gallager_int_stream_simu("run", n=n, wr=wr, wc=wc, max_l=L, num_frames=F, num_cm=ncm,
                                               e_values=e_values, decoder=decoder, seed=5, out_dir=str(tmp_path))
    text = (tmp_path / "run-result.csv").read_text(encoding="utf-8")
    rows, hist = pkg.result_csv.parse_result_csv(text, ncm, len(e_values), L)
    channel, rule = {"LogDecoder": (0, 0), "LogDecoderBefore": (1, 0), "MinSumDecoder": (0, 1)}[decoder]
    for col in range(ncm):
        m, n_, rr, cc = pkg.check_matrix_io.load_check_matrix(str(tmp_path / ("run%d-HMatrix.txt" % col)))
        mm, nn, r0, c0 = pkg.codes.gallager(n, wr, wc, seed=5 + col)
        assert (m, n_) == (mm, nn) and (rr == r0).all() and (cc == c0).all()
        eh, eg = pkg.encoder.systematic_pair(pkg.check_matrix_io.to_dense(m, n_, rr, cc))
        g = oracle.Graph(m, n_, *np.nonzero(eh))
        for i, e in enumerate(e_values):
            c = cnt[col][i]
            assert c[0] == F and hist[col, i].sum() == F
            assert (hist[col, i] == c[16:].astype(np.int64)).all()
            assert abs(float(rows[col][i][3]) - int(c[1]) / F) < 1e-6            # the FER cell
            assert rows[col][i][12] == "(%d/%d)" % (c[7], c[2])
            want = g.simulate(channel, e, ((5 << 16) + col << 8) + i, 0, F, L, cn_rule=rule, G=eg)
            assert c[10] == want[10] and c[11] == want[11]                       # same channel realisation
            assert (c == want).all(), (decoder, col, e, c[:12], want[:12])                 # every counter
    assert cnt[0][1][1] > 0                                                      # the noisy legs do fail


def test_qc_int_stream_simu(pkg, oracle, tmp_path):
    z, mb, nb, L, F = 16, 3, 6, 15, 400
    cnt = pkg.drivers.qc_int_stream_simu("qc", z=z, mb=mb, nb=nb, max_l=L, num_frames=F, num_cm=1, e_values=(0.03,),
                                         decoder="LogDecoderBefore", seed=2, out_dir=str(tmp_path))
    lines = (tmp_path / "qc-result.csv").read_text(encoding="utf-8").split("\n")
    assert lines[1].startswith("%d,%d,%d,0.500000,%d,%d," % (nb * z, nb, mb, L, F))
    m, n, rr, cc = pkg.check_matrix_io.load_check_matrix(str(tmp_path / "qc0-HMatrix.txt"))
    base, (mm, nn, r0, c0) = pkg.codes.random_qc(z, mb, nb, seed=2)
    assert (m, n) == (mb * z, nb * z) and (rr == r0).all() and (cc == c0).all()
    eh, eg = pkg.encoder.systematic_pair(pkg.check_matrix_io.to_dense(m, n, rr, cc))
    want = oracle.Graph(m, n, *np.nonzero(eh)).simulate(1, 0.03, (((2 << 16) + 0) << 8) + 0, 0, F, L, G=eg)
    c = cnt[0][0]
    assert c[0] == F and (c == want).all(), (c[:12], want[:12])


def test_load_int_stream(pkg, tmp_path):
    m, n, r, c = pkg.codes.gallager(512, 8, 2, seed=3)
    pkg.check_matrix_io.save_check_matrix(str(tmp_path / "code-HMatrix.txt"), m, n, r, c)
    cnt = pkg.drivers.load_int_stream("code", 8, 2, max_l=15, num_frames=300, e_values=(0.02,), out_dir=str(tmp_path))
    lines = (tmp_path / "code-result.csv").read_text(encoding="utf-8").split("\n")
    assert lines[1].startswith("512,8,2,0.750000,15,300,") and lines[4].startswith("0.020,")
    assert sum(int(l.split(",")[1]) for l in lines[8:8 + 15]) == 300 == cnt[0][0]
    with pytest.raises(FileNotFoundError):
        pkg.drivers.load_int_stream("missing", 8, 2, out_dir=str(tmp_path))
