"""-m "not gpu": small host formats and the segment-label encoding, pinned with data files the REFERENCE holds
(demo/timit-aux/, copied as fixtures under tests/golden/): the 48-phone symbol list, the phone-duration label map the
demo's scoring tools read, the training output list, the reference's own TIMIT label file (ILAB) and transcripts (MLF).

  * `label = nActualLabs * (dur - 1) + phone` (gradbuilder :216-231; what scrf_viterbi_batch and the lattice olabels
    emit, what `labels` of scrf_utt carries) IS the reference's phn-dur-lab-map-timit.txt, line by line;
  * the checker's label grouping (CRF_InLabStream_SeqMultiWindow restated in oracle/scrf_oracle.c) applied to the
    reference's frame labels yields labels that the map turns back into exactly those frames, long segments split at
    MAXDUR as the map's SPLITLONGSEG says;
  * the ILAB reader finds as many sentences as the reference's training output list names (3696), all phones inside
    the symbol list's alphabet (the label file is a forced alignment: it does not follow the hand transcripts of
    timit_train.mlf frame by frame, so those are not used as a comparator);
  * CRF_MLFManager reads the reference's timit_test39.mlf and gives every transcript back unchanged."""
import os
import subprocess

import numpy as np
import pytest

import orc
from test_qn_files import py_read_ilab

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
L, MAXDUR = 48, 10


def read_symlist():
    lines = open(os.path.join(G, "symlist-timit.txt")).read().split("\n")
    assert int(lines[0]) == L
    sym = dict((n, int(i)) for n, i in (ln.split() for ln in lines[1:] if ln.strip()))
    return sym


def read_label_map():
    head, rows = {}, []
    for ln in open(os.path.join(G, "phn-dur-lab-map-timit.txt")):
        f = ln.split()
        if len(f) == 2:
            head[f[0]] = int(f[1])
        elif len(f) == 3:
            rows.append(tuple(int(v) for v in f))
    return head, rows


def test_symbol_list_is_the_label_alphabet():
    sym = read_symlist()
    assert sorted(sym.values()) == list(range(L)) and sym["sil"] == 0 and sym["zh"] == 47
    sents = py_read_ilab(os.path.join(G, "timit_train.48labs.ilab"))
    assert len(sents) == 3696 and max(max(s) for s in sents) == L - 1 and min(min(s) for s in sents) == 0
    names = [ln.strip() for ln in open(os.path.join(G, "timit_sisx_train.olist")) if ln.strip()]
    assert len(names) == len(sents) and all(n.endswith(".lat") for n in names) and len(set(names)) == len(names)


def test_label_encoding_is_the_references_phone_duration_map():
    head, rows = read_label_map()
    assert head == {"NUML": L * MAXDUR, "SPLITLONGSEG": 1, "MAXDUR": MAXDUR, "GROUPSIZE": 1, "CUTOFF": 0}
    assert len(rows) == L * MAXDUR
    for phone, dur, label in rows:
        assert label == L * (dur - 1) + phone            # gradbuilder :216-231, decoders' olabel - 1
        assert (label % L, label // L + 1) == (phone, dur)   # the decode direction (computeExpF :685-695)


def test_grouped_labels_of_the_reference_label_file_decode_back_through_the_map():
    _, rows = read_label_map()
    back = {label: (phone, dur) for phone, dur, label in rows}
    sents = py_read_ilab(os.path.join(G, "timit_train.48labs.ilab"))
    n_split = 0
    for frames in sents[:300]:
        frames = np.asarray(frames, dtype=np.uint32)
        lab = orc.group_labels(frames, MAXDUR, L)        # one label per frame, 0xffffffff where no segment ends
        rebuilt, t = [], 0
        for e in range(len(frames)):
            if lab[e] == 0xFFFFFFFF:
                continue
            phone, dur = back[int(lab[e])]
            assert e - t + 1 == dur                      # segments tile the utterance
            rebuilt += [phone] * dur
            t = e + 1
        assert rebuilt == list(frames)
        runs = np.diff(np.flatnonzero(np.concatenate(([True], frames[1:] != frames[:-1], [True]))))
        n_split += int((runs > MAXDUR).sum())
    assert n_split > 100                                 # SPLITLONGSEG: long phones do get cut at MAXDUR


def test_mlf_manager_gives_the_references_transcripts_back(tmp_path):
    lib = os.path.join(ROOT, "asr-craft_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libcrf_amd_host.so")):
        import __graft_entry__ as g
        g.build()
    exe = str(tmp_path / "mlf_roundtrip")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "asr-craft_amd", "host"),
                    os.path.join(ROOT, "tests", "host", "mlf_roundtrip.cpp"), "-o", exe, "-L" + lib, "-Wl,-rpath," + lib,
                    "-lcrf_amd_host", "-lscrf_amd"], check=True, timeout=300)
    mlf = os.path.join(G, "timit_test39.mlf")
    r = subprocess.run([exe, mlf, os.path.join(G, "symlist-timit.txt")], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == open(mlf, "rb").read()
    # every utterance of the reference's three test output lists has a transcript there
    keys = set(ln.strip()[1:-5] for ln in open(mlf) if ln.startswith('"'))
    for ol in ("timit_sisx_test.core.neworder.olist", "timit_sisx_test.dt_set.neworder.olist", "timit_sisx_test.core+rest.neworder.olist"):
        names = [ln.strip()[:-4] for ln in open(os.path.join(G, ol)) if ln.strip()]
        assert len(names) > 100 and all(n in keys for n in names), ol


def test_large_weight_file_text_is_the_stream_rendering(tmp_path):
    """The trainer writes three weight-sized text files per epoch, one value per line in the reference's stream format
    (`ofile << lambda[i] << endl`).  Vectors of 65536 values and more are formatted by several threads, slice by slice:
    the file must be the same bytes as the one-value-at-a-time stream rendering (both sizes of the switch)."""
    lib = os.path.join(ROOT, "asr-craft_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libcrf_amd_host.so")):
        import __graft_entry__ as g
        g.build()
    exe = str(tmp_path / "weight_file_text")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "asr-craft_amd", "host"),
                    os.path.join(ROOT, "tests", "host", "weight_file_text.cpp"), "-o", exe, "-L" + lib, "-Wl,-rpath," + lib,
                    "-lcrf_amd_host", "-lscrf_amd"], check=True, timeout=300)
    for n in (1000, 300001):
        r = subprocess.run([exe, str(tmp_path / ("w%d.txt" % n)), str(n)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
