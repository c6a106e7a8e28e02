"""End-to-end run of the reference's TIMIT demo configuration (demo/segmental-timit-demo.cfg.in:
48 phones, max duration 10, ftr1 = 144-dim posteriors as segment features, ftr2 = the same frames
with +-6 frames of context as transition features, stdtrans map, 4 371 216 weights) through the
drop-in front-ends on one MI355X: CRFTrain (AdaGrad SGD, minibatch 64) for a few epochs on binary
pfile + ILAB inputs, then CRFFstDecode and CRFDecode on the first utterances.

The label file is the reference's own demo/timit-aux/timit_train.48labs.ilab (kept as
tests/golden/timit_train.48labs.ilab: 3696 sentences, real TIMIT alignments).  The Kaldi/MLP
posterior features are not redistributable, so they are SYNTHETIC: per frame a softmax over 144
columns of unit Gaussian noise plus `--signal` on the three columns of the frame's phone -- rows
sum to one like the demo's softmax features, and the task is learnable, so the frame accuracy of the
decoded segmentation is a sanity check that training, weight files and decoding fit together.

usage (GPU box): python tools/timit_demo_epoch.py [--utts 3696] [--epochs 2] [--out gpurun_out/timit_demo.json]
"""
import argparse, json, os, struct, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "asr-craft_amd", "bin")
ILAB = os.path.join(ROOT, "tests", "golden", "timit_train.48labs.ilab")


def read_ilab(path):
    d = open(path, "rb").read()
    _, _, idx_off, _, n_sents, _, _ = struct.unpack(">7I", d[4:32])
    idx = struct.unpack(">%dI" % (2 * n_sents), d[idx_off:idx_off + 8 * n_sents])
    out = []
    for s in range(n_sents):
        at, labs = 4 + idx[s], []
        while d[at]:
            c = d[at]
            if c & 0x80:
                c = ((c & 0x7F) << 8) | d[at + 1]
                at += 1
            labs += [d[at + 1]] * c
            at += 2
        out.append(np.asarray(labs, dtype=np.uint32))
    return out


def write_pfile(path, utts):
    W = utts[0].shape[1]
    N = sum(x.shape[0] for x in utts)
    C = 2 + W
    starts = np.zeros(len(utts) + 1, dtype=">u4")
    hdr = ("-pfile_header version 0 size 32768\n-num_sentences %d\n-num_frames %d\n-first_feature_column 2\n-num_features %d\n"
           "-first_label_column %d\n-num_labels 0\n-format dd%s\n-data size %d offset 0 ndim 2 nrow %d ncol %d\n"
           "-sent_table_data size %d offset %d ndim 1\n-end\n" % (len(utts), N, W, C, "f" * W, N * C, N, C, len(utts) + 1, N * C)).encode()
    with open(path, "wb") as f:
        f.write(hdr + b"\0" * (32768 - len(hdr)))
        at = 0
        for u, x in enumerate(utts):
            T = x.shape[0]
            rows = np.empty((T, C), dtype=">u4")
            rows[:, 0] = u
            rows[:, 1] = np.arange(T)
            rows[:, 2:] = x.astype(">f4").view(">u4")
            f.write(rows.tobytes())
            at += T
            starts[u + 1] = at
        f.write(starts.tobytes())


def run(cmd, log):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    open(log, "w").write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    if r.returncode != 0:
        raise SystemExit("%s failed (rc %d), see %s\n%s" % (cmd[0], r.returncode, log, r.stderr[-2000:]))
    return dt, r.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=3696)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--bunch", type=int, default=64)
    ap.add_argument("--eta", type=float, default=0.05)
    ap.add_argument("--signal", type=float, default=2.0)
    ap.add_argument("--decode-utts", type=int, default=200)
    ap.add_argument("--lm-scale", type=float, default=0.1, help="scale of the bigram LM costs (unscaled, -log P per new phone acts as a heavy insertion penalty on top of the CRF's own transition scores)")
    ap.add_argument("--work", default="/tmp/timit_demo")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "timit_demo.json"))
    a = ap.parse_args()
    os.makedirs(a.work, exist_ok=True)
    for fn in os.listdir(a.work):   # a stale .done.train would make CRFTrain return at once (as the reference does)
        if fn.startswith(".done.train") or fn.startswith("weights."):
            os.remove(os.path.join(a.work, fn))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    L, D, W, CTX = 48, 10, 144, 6
    labs = read_ilab(ILAB)[:a.utts]
    U = len(labs)
    rng = np.random.default_rng(1234)
    t0 = time.perf_counter()
    f1, f2 = [], []
    for lab in labs:
        T = lab.shape[0]
        z = rng.standard_normal((T, W)).astype(np.float32)
        for k in range(3):
            z[np.arange(T), 3 * lab + k] += a.signal
        z -= z.max(axis=1, keepdims=True)
        p = np.exp(z)
        p /= p.sum(axis=1, keepdims=True)
        f1.append(p.astype(np.float32))
        f2.append(np.concatenate([np.repeat(p[:1], CTX, 0), p, np.repeat(p[-1:], CTX, 0)]).astype(np.float32))  # feacat -p 6
    pf1, pf2 = os.path.join(a.work, "train.pf"), os.path.join(a.work, "train.pad6.pf")
    write_pfile(pf1, f1)
    write_pfile(pf2, f2)
    t_gen = time.perf_counter() - t0
    n_frames = int(sum(x.shape[0] for x in labs))

    F1 = 8 * W + D
    model = ["crf_label_size=%d" % L, "crf_featuremap=stdtrans", "crf_model_type=stdseg_no_dur_no_segtransftr",
             "label_maximum_duration=%d" % D, "dur_ftr_start=%d" % (8 * W), "num_actual_labs=%d" % L,
             "ftr1_file=" + pf1, "ftr1_window_len=%d" % D, "ftr1_left_context_len=0", "ftr1_right_context_len=0", "ftr1_extract_seg_ftr=1",
             "ftr2_file=" + pf2, "ftr2_window_len=%d" % D, "ftr2_left_context_len=%d" % CTX, "ftr2_right_context_len=%d" % CTX, "ftr2_extract_seg_ftr=0",
             "window_extent=%d" % D, "crf_stateftr_start=0", "crf_stateftr_end=%d" % (F1 - 1), "crf_transftr_start=%d" % F1, "crf_transftr_end=-1"]
    wf = os.path.join(a.work, "weights.48_TIMIT.out")
    t_train, out = run([os.path.join(BIN, "CRFTrain"), "out_weight_file=" + wf, "hardtarget_file=" + ILAB, "train_sent_range=0-%d" % (U - 1),
                        "cv_sent_range=0", "crf_use_adagrad=1", "crf_adagrad_eta=%g" % a.eta, "crf_lr_decay_rate=1.0", "crf_utt_rpt=%d" % (a.bunch * max(1, min(10, (U // a.bunch) // 5))),   # a progress line at least five times per epoch (a count the step never hits -- 640 with 400 utterances -- leaves the trace empty)
                        "crf_states=1", "crf_epochs=%d" % a.epochs, "use_broken_class_label=0", "crf_bunch_size=%d" % a.bunch, "threads=1",
                        "crf_train_order=seq"] + model, os.path.join(a.work, "train.log"))
    lambda_len = int([l for l in out.split("\n") if l.startswith("FEATURES:")][0].split()[1])
    avg_logli = [float(l.split("Iter-Avg LogLi:")[1]) for l in out.split("\n") if "Iter-Avg LogLi:" in l]

    nd = min(a.decode_utts, U)
    hyp = os.path.join(a.work, "hyp.txt")
    t_fst, _ = run([os.path.join(BIN, "CRFFstDecode"), "weight_file=" + wf, "crf_eval_range=0-%d" % (nd - 1), "crf_output_labelfile=" + hyp,
                    "crf_output_format=ascii"] + model, os.path.join(a.work, "fstdecode.log"))
    got = np.loadtxt(hyp, dtype=np.int64).reshape(-1, 3)
    olist, osym, mlf = (os.path.join(a.work, x) for x in ("olist", "osyms.txt", "hyp.mlf"))
    open(olist, "w").write("".join("utt%04d\n" % u for u in range(U)))
    open(osym, "w").write("<eps> 0\n" + "".join("ph%d %d\n" % (i, i + 1) for i in range(L)))
    t_dec, _ = run([os.path.join(BIN, "CRFDecode"), "weight_file=" + wf, "crf_eval_range=0-%d" % (nd - 1), "crf_olist=" + olist, "crf_osymbols=" + osym,
                    "crf_output_mlffile=" + mlf] + model, os.path.join(a.work, "decode.log"))
    # CRFDecode against a phone-bigram LM estimated from the (collapsed) training label sequences, add-one
    # smoothed, as an OpenFST text file: state 0 = start, state p + 1 = last phone p; ilabel = olabel = phone + 1
    big = np.ones((L + 1, L + 1))   # [previous (L = start)][next (L = end)]
    for lab in labs:
        seq = [int(lab[0])] + [int(x) for i, x in enumerate(lab[1:]) if x != lab[i]]
        big[L, seq[0]] += 1
        for a_, b_ in zip(seq[:-1], seq[1:]):
            big[a_, b_] += 1
        big[seq[-1], L] += 1
    big[np.arange(L), np.arange(L)] = 0   # a phone does not follow itself at the LM level (it continues internally)
    big[L, L] = 0
    cost = a.lm_scale * -np.log(big / big.sum(axis=1, keepdims=True), where=big > 0, out=np.full_like(big, np.inf))
    lmf = os.path.join(a.work, "bigram.fst.txt")
    with open(lmf, "w") as fh:
        for l in range(L):
            fh.write("0 %d %d %d %.6f\n" % (l + 1, l + 1, l + 1, cost[L, l]))
        for p_ in range(L):
            for l in range(L):
                if l != p_:
                    fh.write("%d %d %d %d %.6f\n" % (p_ + 1, l + 1, l + 1, l + 1, cost[p_, l]))
        for p_ in range(L):
            fh.write("%d %.6f\n" % (p_ + 1, cost[p_, L]))
    mlf_lm = os.path.join(a.work, "hyp_lm.mlf")
    n_lm = min(nd, 100)
    t_lm, _ = run([os.path.join(BIN, "CRFDecode"), "weight_file=" + wf, "crf_eval_range=0-%d" % (n_lm - 1), "crf_olist=" + olist, "crf_osymbols=" + osym,
                   "crf_output_mlffile=" + mlf_lm, "crf_lm_txt=" + lmf] + model, os.path.join(a.work, "decode_lm.log"))

    def phone_errors(mlf_path, n):
        # Levenshtein distance between the decoded and the reference (collapsed) phone sequences
        err = ref_n = 0
        blocks = open(mlf_path).read().split('"\n')[1:]
        for u in range(n):
            hyp = [int(x[2:]) for x in blocks[u].split("\n") if x.startswith("ph")]
            lab = labs[u]
            ref = [int(lab[0])] + [int(x) for i, x in enumerate(lab[1:]) if x != lab[i]]
            dp = list(range(len(hyp) + 1))
            for i, r_ in enumerate(ref, 1):
                prev, dp[0] = dp[0], i
                for j, h_ in enumerate(hyp, 1):
                    prev, dp[j] = dp[j], min(dp[j] + 1, dp[j - 1] + 1, prev + (r_ != h_))
            err += dp[-1]; ref_n += len(ref)
        return err / max(1, ref_n), sum(len([x for x in b.split('\n') if x.startswith('ph')]) for b in blocks[:n]) / max(1, n)
    (per_free, nph_free), (per_lm, nph_lm) = phone_errors(mlf, n_lm), phone_errors(mlf_lm, n_lm)
    # frame accuracy of the decoded segmentation (CRFFstDecode labels: phone + L * (dur - 1) per segment)
    hit = tot = 0
    phone_seq_equal = 0
    mlf_utts = open(mlf).read().split('"\n')[1:]
    for u in range(nd):
        seg = got[got[:, 0] == u][:, 2]
        fr = np.concatenate([np.full(int(s) // L + 1, int(s) % L) for s in seg]) if len(seg) else np.zeros(0, dtype=np.int64)
        ref = labs[u].astype(np.int64)
        assert fr.shape[0] == ref.shape[0], (u, fr.shape, ref.shape)
        hit += int((fr == ref).sum()); tot += ref.shape[0]
        # CRFDecode's MLF must spell the same phone sequence (repeats of a phone merged)
        phones = [int(s) % L for s in seg]
        merged = [p for i, p in enumerate(phones) if i == 0 or p != phones[i - 1]]
        mlf_ph = [int(x[2:]) for x in mlf_utts[u].split("\n") if x.startswith("ph")]
        phone_seq_equal += int(merged == mlf_ph)
    res = {"workload": "TIMIT demo configuration (48 phones, max duration 10, 144-dim posteriors + +-6-frame context stream, stdtrans), "
                       "real TIMIT label file, synthetic posterior features",
           "utterances": U, "frames": n_frames, "lambda_len": lambda_len, "epochs": a.epochs, "minibatch": a.bunch, "adagrad_eta": a.eta,
           "feature_generation_s": round(t_gen, 1),
           "crftrain_wall_s": round(t_train, 2), "crftrain_utt_per_s_incl_file_io": round(U * a.epochs / t_train, 1),
           "iter_avg_logli_trace": avg_logli[:3] + avg_logli[-3:],
           "crffstdecode_wall_s": round(t_fst, 2), "crfdecode_wall_s": round(t_dec, 2), "decoded_utterances": nd,
           "decoded_frame_accuracy": round(hit / max(1, tot), 4), "crfdecode_mlf_equals_crffstdecode_phones": "%d/%d" % (phone_seq_equal, nd),
           "crfdecode_bigram_lm": {"utterances": n_lm, "lm_scale": a.lm_scale, "wall_s": round(t_lm, 2), "phone_error_rate_free_loop": round(per_free, 4),
                                   "phone_error_rate_bigram_lm": round(per_lm, 4), "phones_per_utt_free_loop": round(nph_free, 1), "phones_per_utt_bigram_lm": round(nph_lm, 1)}}
    open(a.out, "w").write(json.dumps(res, indent=1) + "\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
