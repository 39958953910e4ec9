"""Several builds at the same time on one GPU: one context (DeviceBuilder) per host thread, each with its own stream, scalars and
mapped read-back buffer (Pipeline::sync_reads spins on a word in mapped pinned memory; a repeat-rich build adds a helper pipeline
on a thread of its own).  Every build equals the oracle; the library's contract is one thread per context, any number of contexts
(INTEGRATION.md)."""
import threading

import numpy as np
import pytest
import torch

import sufr_amd
from sufr_amd import synth

pytestmark = pytest.mark.gpu


def _texts():
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    a = acgt[rng.integers(0, 4, 300_000)].copy(); a[-1] = ord("$")
    out.append((a, dict(is_dna=True)))
    b = acgt[rng.choice(4, 2_000_000, p=[.5, .2, .2, .1])].copy()          # repeat-rich: planted copies, a homopolymer run
    fam = acgt[rng.integers(0, 4, 3000)]
    for at in rng.integers(0, b.size - 3000, 200): b[at:at + 3000] = fam
    b[1_000_000:1_100_000] = ord("A"); b[-1] = ord("$")
    out.append((b, dict(is_dna=True)))
    c, _ = synth.syn_human(1_500_000, seed=9)
    out.append((c.numpy().copy(), dict(is_dna=True, ignore_softmask=True)))
    d = a.copy(); d[rng.integers(0, d.size - 1, 30)] = ord("R")            # listed bytes: the exceptions path
    out.append((d, dict(is_dna=True)))
    return out


def test_four_contexts_on_four_threads_equal_the_oracle(oracle):
    cases = _texts()
    want = []
    for raw, kw in cases:
        osa, olcp, _ = oracle.build(oracle.normalize(raw, kw.get("ignore_softmask", False)), is_dna=True, threads=8)
        want.append((osa, olcp))
    errors = []

    def work(k):
        try:
            raw, kw = cases[k]
            x = torch.from_numpy(raw).cuda()
            db = sufr_amd.DeviceBuilder(0)
            for rep in range(6):
                sa, lcp = db.sort(x, raw_text=True, **kw)
                gsa = sa.cpu().numpy().view(np.uint32); glcp = lcp.cpu().numpy().view(np.uint32)
                if not (np.array_equal(gsa, want[k][0]) and np.array_equal(glcp, want[k][1])):
                    errors.append(f"text {k}, repetition {rep}: arrays differ from the oracle's")
                    break
            db.close()
        except Exception as e:                        # noqa: BLE001 (reported below, in the main thread)
            errors.append(f"text {k}: {e!r}")

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(cases))]
    for t in threads: t.start()
    for t in threads: t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a build did not return"
    assert not errors, errors
