"""Fuzz loop over damaged inputs against the HOST-ONLY sanitizer build of the product's host code
(sufr_amd/csrc `make asan`: sufr_io.cpp + sufr_query.cpp under AddressSanitizer + UBSan, device entry points stubbed).

Run by tests/test_sanitized_host.py (and profiles/asan_host.sh) in a child process:
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) SUFR_AMD_HOST_ASAN_LIB=1 python tests/fuzz_host.py [iterations] [seed]

Inputs: the reference's own test inputs (tests/golden/inputs/*.fa, a FASTQ made from one of them) plain and as gzip /
bzip2 / xz streams, and its golden .sufr files (tests/golden/expected/*.sufr).  Every iteration damages one of them --
truncation, bit flips (dense in the first bytes, where the headers are), overwritten runs, appended garbage, an empty file
-- and hands it to the reader (sufr_read_sequence_file, util.rs:51-89) or to the .sufr parser (sufr_file_open,
sufr_file.rs:145-275) and, when that accepts the file, to every accessor and to the host search.
The contract: each call returns -- with data, or with an error code AND a message.  A crash, a sanitizer report (the
process aborts) or an error without a message fails the run.
"""
import bz2
import ctypes as C
import gzip
import lzma
import os
import random
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import sufr_amd                                             # noqa: E402
from sufr_amd._lib import FileMeta, SequenceData            # noqa: E402

GOLDEN = ROOT / "tests" / "golden"


def damage(rng: random.Random, data: bytes) -> bytes:
    b = bytearray(data)
    kind = rng.randrange(8)
    if kind == 0 or not b:
        return bytes(b[:rng.randrange(len(b) + 1)])                       # truncation (down to an empty file)
    if kind == 1:                                                         # bit flips in the first bytes: magic numbers, headers
        for _ in range(rng.randrange(1, 6)):
            i = rng.randrange(min(len(b), 96)); b[i] ^= 1 << rng.randrange(8)
    elif kind == 2:                                                       # bit flips anywhere
        for _ in range(rng.randrange(1, 12)):
            i = rng.randrange(len(b)); b[i] ^= 1 << rng.randrange(8)
    elif kind == 3:                                                       # a run overwritten with one byte
        i = rng.randrange(len(b)); n = rng.randrange(1, 64)
        b[i:i + n] = bytes([rng.choice([0, 0xFF, 0x0A, 0x3E, 0x40, 0x2B, rng.randrange(256)])]) * min(n, len(b) - i)
    elif kind == 4:                                                       # a header integer replaced by an extreme value
        i = rng.randrange(0, max(1, min(len(b) - 8, 90)))
        v = rng.choice([0, 1, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 1 << 63, len(b), len(b) + 1, len(b) * 8, rng.getrandbits(64)])
        b[i:i + 8] = (v & 0xFFFFFFFFFFFFFFFF).to_bytes(8, "little")
    elif kind == 5:
        b += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 200)))      # garbage appended
    elif kind == 6:                                                       # truncated AND flipped
        del b[rng.randrange(len(b)):]
        if b:
            i = rng.randrange(len(b)); b[i] ^= 1 << rng.randrange(8)
    else:                                                                 # a slice cut out of the middle
        i = rng.randrange(len(b)); j = min(len(b), i + rng.randrange(1, 400))
        del b[i:j]
    return bytes(b)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    L = sufr_amd.lib()
    assert sufr_amd._lib.HOST_ASAN, "run with SUFR_AMD_HOST_ASAN_LIB=1 (and libasan preloaded)"

    fastas = sorted((GOLDEN / "inputs").glob("*.fa"))
    seq_seeds = []
    for f in fastas:
        raw = f.read_bytes()
        if len(raw) > 200_000:
            raw = raw[:200_000]
        seq_seeds.append((f.name, raw))
    fq = b"".join(b"@r%d some text\nACGTNACGTTGCA%s\n+\nIIIIIIIIIIIII%s\n" % (i, b"ACGT" * i, b"IIII" * i) for i in range(40))
    seq_seeds.append(("reads.fq", fq))
    packed = []
    for name, raw in seq_seeds:
        packed.append((name, raw))
        packed.append((name + ".gz", gzip.compress(raw, 6)))
        packed.append((name + ".bz2", bz2.compress(raw)))
        packed.append((name + ".xz", lzma.compress(raw)))
    sufrs = [(f.name, f.read_bytes()) for f in sorted((GOLDEN / "expected").glob("*.sufr")) if f.stat().st_size < 700_000]

    tmp = Path(tempfile.mkdtemp(prefix="sufr_fuzz_"))
    stats = {"seq_ok": 0, "seq_err": 0, "sufr_ok": 0, "sufr_err": 0, "searches": 0}
    queries = [b"A", b"AC", b"ACGT", b"GATTACA", b"NNNN", b"M", b"MKV", b"$", b"", b"ACGTACGTACGTACGTACGTACGT", bytes(range(1, 40))]
    for it in range(iters):
        if it % 2 == 0:
            name, data = packed[rng.randrange(len(packed))]
            path = tmp / ("in_" + name)
            path.write_bytes(damage(rng, data))
            sd = SequenceData()
            err = C.create_string_buffer(512)
            rc = L.sufr_read_sequence_file(os.fsencode(str(path)), ord("%"), C.byref(sd), err, len(err))
            if rc == 0:
                stats["seq_ok"] += 1
                text = C.string_at(sd.seq, sd.seq_len) if sd.seq_len else b""
                assert sd.seq_len >= 1 and text[-1:] == b"$", "an accepted file yields a '$'-terminated text"
                for i in range(sd.num_sequences):                        # every start inside the text, every name a C string
                    assert sd.start_positions[i] < sd.seq_len
                    _ = sd.sequence_names[i]
                L.sufr_sequence_data_free(C.byref(sd))
            else:
                stats["seq_err"] += 1
                assert err.value, f"error {rc} without a message for {name}"
        else:
            name, data = sufrs[rng.randrange(len(sufrs))]
            path = tmp / ("f_" + name)
            path.write_bytes(damage(rng, data))
            h = C.c_void_p()
            err = C.create_string_buffer(512)
            rc = L.sufr_file_open(str(path).encode(), C.byref(h), err, len(err))
            if rc != 0:
                stats["sufr_err"] += 1
                assert err.value, f"error {rc} without a message for {name}"
                continue
            stats["sufr_ok"] += 1
            m = FileMeta()
            L.sufr_file_metadata(h, C.byref(m))
            n, s = m.text_len, m.len_suffixes
            w = m.index_width
            if n:
                _ = C.string_at(L.sufr_file_text(h), n)                  # the whole text section is readable
            if s:
                _ = C.string_at(L.sufr_file_suffix_array(h), s * w)
                _ = C.string_at(L.sufr_file_lcp_array(h), s * w)
                for r in (0, s // 2, s - 1):
                    L.sufr_file_suffix(h, r); L.sufr_file_lcp(h, r)
            if m.seed_mask_len:
                _ = C.string_at(L.sufr_file_seed_mask(h), m.seed_mask_len)
            for i in range(min(m.num_sequences, 50)):
                L.sufr_file_sequence_start(h, i)
                _ = L.sufr_file_sequence_name(h, i)
            for p in (0, n // 2, n - 1 if n else 0, n, n + 7, 1 << 62):
                L.sufr_file_sequence_of(h, p)
            lo, hi = C.c_uint64(), C.c_uint64()
            for q in queries:                                            # a damaged SA / LCP section must not take the search out of the text
                for mql in (None, 1, 3, 1000):
                    L.sufr_file_search(h, q, len(q), int(mql is not None), mql or 0, C.byref(lo), C.byref(hi))
                    stats["searches"] += 1
            L.sufr_file_close(h)
    for f in tmp.iterdir():
        f.unlink()
    tmp.rmdir()
    print(f"fuzz_host: {iters} damaged inputs (seed {seed}): reader accepted {stats['seq_ok']} / refused {stats['seq_err']} with a "
          f"message; .sufr parser accepted {stats['sufr_ok']} / refused {stats['sufr_err']} with a message; {stats['searches']} "
          f"host searches on accepted files; no crash, no sanitizer report")


if __name__ == "__main__":
    main()
