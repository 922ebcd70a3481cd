"""GlobalChain over the priority search tree (GlobalChain.h:85-189, PrioritySearchTree.h): the oracle's restatement is PINNED to the reference's own templates
(tests/golden/globalchain_golden.json, made by tools/make_golden_globalchain.py from oracle/ref_harness/globalchain_ref.cpp; case 0 is the input of the
reference's TestGlobalChain.cpp), and the HIP path is compared with both."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "globalchain_golden.json")


def _cases():
    return json.load(open(GOLD))["cases"]


def test_oracle_global_chain_matches_reference(oracle):
    cases = _cases()
    # the known answer of TestGlobalChain.cpp: the eight collinear fragments, the off-diagonal one left out
    assert cases[0]["chain"] == list(range(8)) and [cases[0]["score"][c] for c in cases[0]["chain"]] == [10, 20, 30, 40, 50, 60, 70, 80]
    n_long = 0
    for i, c in enumerate(cases):
        ch, sc, pv = O.global_chain(c["fragments"])
        assert ch == c["chain"] and sc == c["score"] and pv == c["prev"], i
        n_long += len(ch) >= 5
    assert len(cases) == 200 and n_long >= 50


@pytest.mark.gpu
def test_hip_global_chain(ctx, oracle):
    import torch
    from lra_amd import chain
    cases = _cases()
    rng = np.random.default_rng(3)
    extra = []
    for _ in range(300):                                                  # larger sets than the golden file holds
        n = int(rng.integers(1, 400)); g = int(rng.choice([1, 3, 20]))
        x = np.cumsum(rng.integers(0, 8, n)) * g; y = np.cumsum(rng.integers(0, 8, n)) * g + rng.integers(-3, 4, n) * g * (rng.random(n) < 0.2)
        ln = rng.integers(1, 10, n) * g
        fr = np.stack([x, np.maximum(y, 0), x + ln, np.maximum(y, 0) + ln], 1)[rng.permutation(n)]
        extra.append(fr.tolist())
    sets = [c["fragments"] for c in cases] + extra
    off = np.concatenate([[0], np.cumsum([len(s) for s in sets])]).astype(np.int64)
    allf = np.concatenate([np.asarray(s, np.int32).reshape(-1, 4) for s in sets])
    dev = ctx.device
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    res = chain.global_chain_batch(ctx, tt(off), tt(allf[:, 0]), tt(allf[:, 1]), tt(allf[:, 2]), tt(allf[:, 3]), tt((allf[:, 2] - allf[:, 0]).astype(np.int32)))
    nf = int(off[-1])
    score = ctx.to_host(res.d_score, nf, np.int32); prev = ctx.to_host(res.d_prev, nf, np.int32); ch = ctx.to_host(res.d_chain, nf, np.int32)
    clen = ctx.to_host(res.d_chain_len, len(sets), np.uint32)
    for i, s in enumerate(sets):
        a, b = int(off[i]), int(off[i + 1])
        if i < len(cases):
            exp = (cases[i]["chain"], cases[i]["score"], cases[i]["prev"])
        else:
            exp = O.global_chain(s)
        assert ch[a:a + int(clen[i])].tolist() == exp[0] and score[a:b].tolist() == exp[1] and prev[a:b].tolist() == exp[2], i
