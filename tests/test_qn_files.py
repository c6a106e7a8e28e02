"""-m "not gpu": the pfile / ILAB readers and writers of asr-craft_amd/host/qn_files.h (SURVEY
row f1), driven through the host-only `qn_filetool`, against independent Python parsers of the
same byte layouts and against the reference's own ILAB data file
(tests/golden/timit_train.48labs.ilab = demo/timit-aux/timit_train.48labs.ilab)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
TOOL = os.path.join(ROOT, "asr-craft_amd", "bin", "qn_filetool")
ILAB = os.path.join(G, "timit_train.48labs.ilab")


@pytest.fixture(scope="module", autouse=True)
def _tool():
    if not os.path.exists(TOOL):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "asr-craft_amd", "host"), TOOL])


def run(*args, ok=True):
    r = subprocess.run([TOOL] + [str(a) for a in args], capture_output=True, text=True, timeout=120)
    if ok:
        assert r.returncode == 0, r.stdout + r.stderr
    return r


def py_read_ilab(path):
    """Independent parser of the layout documented in qn_files.h."""
    d = open(path, "rb").read()
    assert d[:4] == b"ILAB"
    version, hdr, idx_off, bits, n_sents, n_frames, zero = struct.unpack(">7I", d[4:32])
    assert (version, hdr, bits, zero) == (19990304, 28, 8, 0)
    idx = struct.unpack(">%dI" % (2 * n_sents), d[idx_off:idx_off + 8 * n_sents])
    assert idx_off + 8 * n_sents == len(d)
    offs, cnts = idx[:n_sents], idx[n_sents:]
    out = []
    for s in range(n_sents):
        at = 4 + offs[s]
        labs = []
        while True:
            c = d[at]
            if c == 0:
                break
            if c & 0x80:
                c = ((c & 0x7F) << 8) | d[at + 1]
                at += 1
            labs += [d[at + 1]] * c
            at += 2
        assert len(labs) == cnts[s]
        out.append(labs)
    assert sum(cnts) == n_frames
    return out


def py_write_pfile(path, utts, labs=None):
    """Independent writer of the pfile(5) layout: utts[u] = [T][W] float32, labs[u] = [T][NL]."""
    W = utts[0].shape[1]
    NL = 0 if labs is None else labs[0].shape[1]
    rows, starts = [], [0]
    for u, x in enumerate(utts):
        for t in range(x.shape[0]):
            rows.append(struct.pack(">2i", u, t) + x[t].astype(">f4").tobytes() + (labs[u][t].astype(">i4").tobytes() if NL else b""))
        starts.append(starts[-1] + x.shape[0])
    N, C = starts[-1], 2 + W + NL
    hdr = ("-pfile_header version 0 size 32768\n-num_sentences %d\n-num_frames %d\n-first_feature_column 2\n-num_features %d\n"
           "-first_label_column %d\n-num_labels %d\n-format dd%s%s\n-data size %d offset 0 ndim 2 nrow %d ncol %d\n"
           "-sent_table_data size %d offset %d ndim 1\n-end\n" % (len(utts), N, W, 2 + W, NL, "f" * W, "d" * NL, N * C, N, C, len(utts) + 1, N * C)).encode()
    with open(path, "wb") as f:
        f.write(hdr + b"\0" * (32768 - len(hdr)))
        f.write(b"".join(rows))
        f.write(struct.pack(">%di" % len(starts), *starts))


def py_read_pfile(path):
    d = open(path, "rb").read()
    head = d[:32768].split(b"\0")[0].decode().split("\n")
    kv = {l.split()[0]: l.split()[1:] for l in head if l.strip()}
    S, N, W, NL = (int(kv[k][0]) for k in ("-num_sentences", "-num_frames", "-num_features", "-num_labels"))
    C = 2 + W + NL
    body = np.frombuffer(d[32768:32768 + 4 * N * C], dtype=">u4").reshape(N, C)
    starts = np.frombuffer(d[32768 + 4 * N * C:32768 + 4 * N * C + 4 * (S + 1)], dtype=">u4")
    assert len(d) == 32768 + 4 * N * C + 4 * (S + 1)
    ftr = body[:, 2:2 + W].astype("<u4").view("<f4")
    return [(ftr[starts[s]:starts[s + 1]], body[starts[s]:starts[s + 1], 2 + W:].astype(np.uint32)) for s in range(S)], body[:, :2]


def test_ilab_reader_on_the_reference_label_file(tmp_path):
    want = py_read_ilab(ILAB)
    assert len(want) == 3696 and sum(len(x) for x in want) == 1124823 and max(max(x) for x in want) == 47
    out = tmp_path / "labs.ascii"
    r = run("ilab2ascii", ILAB, out)
    assert "sentences 3696 frames 1124823 label_bits 8" in r.stdout
    got = np.loadtxt(out, dtype=np.int64)
    flat = np.concatenate([np.asarray(x) for x in want])
    assert got.shape == (1124823, 3) and np.array_equal(got[:, 2], flat)
    sent = np.concatenate([np.full(len(x), s) for s, x in enumerate(want)])
    frame = np.concatenate([np.arange(len(x)) for x in want])
    assert np.array_equal(got[:, 0], sent) and np.array_equal(got[:, 1], frame)


def test_ilab_writer_reproduces_the_reference_file_byte_for_byte(tmp_path):
    asc, out = tmp_path / "labs.ascii", tmp_path / "again.ilab"
    run("ilab2ascii", ILAB, asc)
    run("ascii2ilab", asc, out)
    assert open(out, "rb").read() == open(ILAB, "rb").read()


def test_ilab_run_lengths_and_edges(tmp_path):
    # runs of 1, 127, 128 (two-byte count), 32767 and 32768 (split), an empty sentence, one sentence
    sents = [[5] * 1 + [6] * 127 + [7] * 128, [], [9] * 32767 + [9] + [3] * 2, [255, 0, 255]]
    asc = tmp_path / "in.ascii"
    with open(asc, "w") as f:
        for s, labs in enumerate(sents):
            for t, l in enumerate(labs):
                f.write("%d %d %d\n" % (s, t, l))
    out = tmp_path / "x.ilab"
    run("ascii2ilab", asc, out)
    assert py_read_ilab(out) == sents
    d = open(out, "rb").read()
    assert d[32:32 + 7] == bytes([1, 5, 127, 6, 0x80, 128, 7])  # 128 takes the two-byte count
    back = tmp_path / "back.ascii"
    run("ilab2ascii", out, back)
    assert open(back).read() == open(asc).read()


@pytest.mark.parametrize("damage", ["magic", "version", "bits", "truncate", "terminator", "count", "total"])
def test_ilab_reader_refuses_damaged_files(tmp_path, damage):
    asc = tmp_path / "in.ascii"
    with open(asc, "w") as f:
        for s in range(3):
            for t in range(5 + s):
                f.write("%d %d %d\n" % (s, t, (s + t // 2) % 7))
    good = tmp_path / "good.ilab"
    run("ascii2ilab", asc, good)
    d = bytearray(open(good, "rb").read())
    if damage == "magic": d[0:4] = b"ILAX"
    elif damage == "version": d[4:8] = struct.pack(">I", 20200101)
    elif damage == "bits": d[16:20] = struct.pack(">I", 16)
    elif damage == "truncate": d = d[:-3]
    elif damage == "terminator": d[d.index(b"\0\0\0\0\1", 32) + 4] = 7
    elif damage == "count": d[-1] += 1
    elif damage == "total": d[24:28] = struct.pack(">I", 99)
    bad = tmp_path / "bad.ilab"
    open(bad, "wb").write(bytes(d))
    r = run("ilab2ascii", bad, tmp_path / "o.ascii", ok=False)
    assert r.returncode == 1 and "Exception:" in r.stderr


def test_pfile_reader_against_an_independent_writer(tmp_path):
    rng = np.random.default_rng(7)
    utts = [rng.standard_normal((T, 5)).astype(np.float32) for T in (3, 1, 8)]
    utts[0][0, 0], utts[0][0, 1], utts[2][7, 4] = np.float32(1e-42), np.float32(-0.0), np.float32(3.4e38)  # denormal, -0, near max
    labs = [rng.integers(0, 1000, (x.shape[0], 2)).astype(np.uint32) for x in utts]
    pf = tmp_path / "a.pfile"
    py_write_pfile(pf, utts, labs)
    r = run("pfile_info", pf)
    assert r.stdout.split() == ["sentences", "3", "frames", "12", "features", "5", "labels", "2"]
    asc = tmp_path / "a.ascii"
    run("pfile2ascii", pf, asc)
    got = np.loadtxt(asc)
    want = np.concatenate([np.concatenate([np.full((x.shape[0], 1), u), np.arange(x.shape[0])[:, None], x.astype(np.float64), labs[u]], axis=1) for u, x in enumerate(utts)])
    assert got.shape == want.shape
    assert np.array_equal(got[:, 2:7].astype(np.float32).view(np.uint32), want[:, 2:7].astype(np.float32).view(np.uint32))  # %.9g round-trips float32 bit for bit
    assert np.array_equal(got[:, [0, 1, 7, 8]], want[:, [0, 1, 7, 8]])


def test_pfile_writer_against_an_independent_reader_and_round_trip(tmp_path):
    rng = np.random.default_rng(8)
    asc = tmp_path / "in.ascii"
    utts = [rng.standard_normal((T, 4)).astype(np.float32) for T in (6, 2, 1, 9)]
    with open(asc, "w") as f:
        for u, x in enumerate(utts):
            for t in range(x.shape[0]):
                f.write("%d %d %s %d\n" % (u, t, " ".join("%.9g" % v for v in x[t]), (u * 7 + t) % 48))
    pf = tmp_path / "out.pfile"
    run("ascii2pfile", asc, pf, 1)
    sents, cols = py_read_pfile(pf)
    assert len(sents) == 4
    for u, (f, l) in enumerate(sents):
        assert np.array_equal(f.view(np.uint32), utts[u].view(np.uint32))
        assert np.array_equal(l[:, 0], (u * 7 + np.arange(utts[u].shape[0])) % 48)
    assert np.array_equal(cols[:, 0], np.concatenate([np.full(x.shape[0], u) for u, x in enumerate(utts)]))
    assert np.array_equal(cols[:, 1], np.concatenate([np.arange(x.shape[0]) for x in utts]))
    head = open(pf, "rb").read(32768)
    assert head.startswith(b"-pfile_header version 0 size 32768\n") and b"\n-format ddffffd\n" in head and b"\n-end\n" in head
    back = tmp_path / "back.ascii"
    run("pfile2ascii", pf, back)
    assert np.array_equal(np.loadtxt(back), np.loadtxt(asc))


def test_pfile_without_sentence_index_and_damaged_pfiles(tmp_path):
    rng = np.random.default_rng(9)
    utts = [rng.standard_normal((T, 3)).astype(np.float32) for T in (4, 2)]
    pf = tmp_path / "a.pfile"
    py_write_pfile(pf, utts)
    d = bytearray(open(pf, "rb").read())
    # drop the -sent_table_data line: the reader rebuilds the index from the sentence column
    head = bytes(d[:32768]).split(b"\0")[0]
    lines = [l for l in head.split(b"\n") if not l.startswith(b"-sent_table_data")]
    nh = b"\n".join(lines)
    noidx = tmp_path / "noidx.pfile"
    open(noidx, "wb").write(nh + b"\0" * (32768 - len(nh)) + bytes(d[32768:32768 + 4 * 6 * 5]))
    a, b = tmp_path / "a.ascii", tmp_path / "b.ascii"
    run("pfile2ascii", pf, a)
    run("pfile2ascii", noidx, b)
    assert open(a).read() == open(b).read()
    for name, mut in [("short", lambda x: x[:-40]), ("nothdr", lambda x: b"-xfile" + x[6:]),
                      ("frameno", lambda x: x[:32768 + 4] + struct.pack(">i", 5) + x[32768 + 8:])]:
        bad = tmp_path / (name + ".pfile")
        open(bad, "wb").write(mut(bytes(d)))
        r = run("pfile2ascii", bad, tmp_path / "o.ascii", ok=False)
        assert r.returncode == 1 and "Exception:" in r.stderr, name


@pytest.mark.parametrize("spec,n,want", [
    ("all", 4, [0, 1, 2, 3]), ("nil", 4, []), ("2", 4, [2]), ("0:2", 4, [0, 1, 2]), ("1-3", 5, [1, 2, 3]),
    ("0:2:6", 8, [0, 2, 4, 6]), ("^0", 5, [4]), ("^2:^0", 5, [2, 3, 4]), ("0,2 3", 4, [0, 2, 3]), ("0:9,20:29", 40, list(range(10)) + list(range(20, 30)))])
def test_sentence_ranges(spec, n, want):
    assert [int(x) for x in run("range", spec, n).stdout.split()] == want


@pytest.mark.parametrize("spec", ["0:9", "x", "3:1:", "1:0:4", "-1"])
def test_sentence_range_errors(spec):
    r = run("range", spec, 4, ok=False)
    assert r.returncode == 1 and "Exception:" in r.stderr


def test_readers_under_address_and_ub_sanitizers_on_damaged_files(tmp_path):
    """CPU sanitizer run (GPU sanitizers are not available on the pool): qn_filetool built with
    -fsanitize=address,undefined reads 300 randomly damaged ILAB / pfile inputs -- every one is either
    rejected with a message or read, none produces a sanitizer report; the reference's label file
    survives the round trip under the sanitizers too."""
    import random
    exe = str(tmp_path / "qn_asan")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", exe,
                        os.path.join(ROOT, "asr-craft_amd", "host", "qn_filetool.cpp")], capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime in this image: " + r.stderr[-200:])

    def tool(*args):
        p = subprocess.run([exe] + [str(a) for a in args], capture_output=True, timeout=120)
        err = p.stderr.decode("utf-8", "replace")
        assert "AddressSanitizer" not in err and "runtime error" not in err, err[:3000]
        return p.returncode

    assert tool("ilab2ascii", ILAB, tmp_path / "l.ascii") == 0 and tool("ascii2ilab", tmp_path / "l.ascii", tmp_path / "l.ilab") == 0
    assert open(tmp_path / "l.ilab", "rb").read() == open(ILAB, "rb").read()
    rnd = random.Random(5)
    with open(tmp_path / "s.ascii", "w") as f:
        f.write("".join("%d %d %d\n" % (s, t, (s * 3 + t // 4) % 48) for s in range(5) for t in range(30 + s)))
    with open(tmp_path / "f.ascii", "w") as f:
        f.write("".join("%d %d %s\n" % (s, t, " ".join("%.3f" % rnd.random() for _ in range(7))) for s in range(4) for t in range(9 + s)))
    assert tool("ascii2ilab", tmp_path / "s.ascii", tmp_path / "s.ilab") == 0 and tool("ascii2pfile", tmp_path / "f.ascii", tmp_path / "f.pfile") == 0
    rejected = 0
    for name, cmd in (("s.ilab", "ilab2ascii"), ("f.pfile", "pfile2ascii")):
        good = open(tmp_path / name, "rb").read()
        for _ in range(150):
            b = bytearray(good)
            kind = rnd.choice(["flip", "trunc", "zero"])
            if kind == "flip":
                for _k in range(rnd.randint(1, 4)):
                    b[rnd.randrange(len(b)) if name == "s.ilab" or rnd.random() < 0.5 else rnd.randrange(400)] = rnd.randrange(256)
            elif kind == "trunc":
                b = b[:rnd.randrange(len(b))]
            else:
                i = rnd.randrange(len(b)); b[i:i + 8] = bytes(8)
            open(tmp_path / "m.bin", "wb").write(bytes(b))
            rejected += tool(cmd, tmp_path / "m.bin", tmp_path / "o.ascii") != 0
    assert rejected > 100
