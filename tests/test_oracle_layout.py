"""a6: lambda index layout (ftrmaps/CRF_StdFeatureMap.cpp:280-517).  Known answers:
lambda_len values the survey's harness measured on the compiled reference
(SURVEY.md section 6 / BASELINE.md section 2) and the demo's 4,371,216 weights."""
import numpy as np
import pytest

import orc


@pytest.mark.parametrize("kw,expect", [
    # cfg 2: L=48 D=25 in=39 stdstate                (harness-verified 18,528)
    (dict(L=48, D=25, F=8 * 39 + 25), 18528),
    # L=48 D=10 in=144 stdstate                      (58,128)
    (dict(L=48, D=10, F=8 * 144 + 10), 58128),
    # same + 144 trans ftrs                          (389,904)
    (dict(L=48, D=10, F=8 * 144 + 10 + 144, sfe=1161, use_trans_ftrs=True, tfs=1162), 389904),
    # frame CRF L=48 in=39 stdtrans                  (94,080)
    (dict(model_type=orc.STDFRAME, L=48, D=1, F=39, use_trans_ftrs=True), 94080),
    # TIMIT demo: 1162 state + 1872 trans ftrs       (4,371,216; cfg.in:2,13-33)
    (dict(L=48, D=10, F=1162 + 1872, sfe=1161, use_trans_ftrs=True, tfs=1162), 4371216),
    # cfg 1: frame CRF, 6 joined ftrs, stdstate      (2,640)
    (dict(model_type=orc.STDFRAME, L=48, D=1, F=6), 2640),
    # cfg 5 stress                                   (245,000)
    (dict(L=200, D=40, F=8 * 123 + 40), 245000),
])
def test_lambda_len_known_answers(kw, expect):
    lay = orc.Layout(orc.config(**kw))
    assert lay.lambda_len == expect


@pytest.mark.parametrize("kw", [
    dict(L=3, D=2, F=5), dict(L=4, D=3, F=7, sfe=3, use_trans_ftrs=True, tfs=4),
    dict(L=5, D=1, F=3, use_state_bias=False), dict(L=2, D=2, F=4, use_trans_bias=False, use_trans_ftrs=True),
])
def test_blocks_partition_lambda(kw):
    cfg = orc.config(**kw)
    lay = orc.Layout(cfg)
    L = cfg.num_labs
    seen = np.zeros(lay.lambda_len, dtype=int)
    for c in range(L):
        seen[lay.state_idx[c]:lay.state_idx[c] + lay.num_state_funcs] += 1
        for p in range(L):
            s = lay.trans_idx[p * L + c]
            seen[s:s + lay.num_trans_funcs] += 1
    assert (seen == 1).all()
    # per-label block: [state funcs][for p: trans funcs (p->c)]
    stride = lay.num_state_funcs + L * lay.num_trans_funcs
    assert (lay.state_idx == np.arange(L) * stride).all()
