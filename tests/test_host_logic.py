"""CPU-side checks of the product's host layer (no GPU, no compute calls):
the C ABI loads and exports every symbol include/sufr_hip.h declares; the sequence-file reader, the text
map, the .sufr writer and lcp_pair agree with the oracle / the golden files; the build entry points fail
loudly when no GPU is present (there is no CPU fallback)."""
import ctypes as C
import re
import subprocess

import numpy as np
import pytest
import torch

import sufr_amd
from oracle_helper import GOLDEN, GOLDEN_CASES, parse_sufr

HAS_GPU = torch.cuda.is_available()


def test_library_is_built_and_exports_every_declared_symbol():
    hdr = (sufr_amd.LIB_PATH.parents[3] / "include" / "sufr_hip.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sufr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(sufr_amd.EXPORTS), declared ^ set(sufr_amd.EXPORTS)
    L = sufr_amd.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.sufr_hip_abi_version() == 3
    # the shared object itself (not just the ctypes table) exports them
    nm = subprocess.run(["nm", "-D", "--defined-only", str(sufr_amd.LIB_PATH)], capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(rf"\bT {name}\b", nm), name


def test_product_does_not_reference_the_oracle():
    root = sufr_amd.LIB_PATH.parents[3]
    for p in list((root / "sufr_amd").rglob("*.py")) + list((root / "sufr_amd" / "csrc").glob("*.*")):
        if p.suffix in {".py", ".hip", ".cpp", ".inc", ".h"}:
            assert "oracle" not in p.read_text().lower().replace("oracle_helper", ""), p


@pytest.mark.parametrize("fa", ["1.fa", "2.fa", "3.fa", "abba.fa", "smol.fa", "long_dna_sequence.fa", "uniprot.fa",
                                "mostlya1.fa", "mostlya2.fa", "spaced_input.fa"])
@pytest.mark.parametrize("delim", [b"%", b"N"])
def test_read_sequence_file_matches_oracle(oracle, fa, delim):
    want = oracle.read_sequence_file(GOLDEN / "inputs" / fa, delim)
    got = sufr_amd.read_sequence_file(GOLDEN / "inputs" / fa, delim)
    assert (got.seq, got.start_positions, got.sequence_names) == want


def test_read_sequence_file_kat():  # util.rs:184-194
    d = sufr_amd.read_sequence_file(GOLDEN / "inputs" / "2.fa", b"N")
    assert d.seq == b"ACGTacgtNacgtACGT$" and d.start_positions == [0, 9] and d.sequence_names == ["ABC", "DEF"]


def test_read_sequence_file_variants(tmp_path):
    p = tmp_path / "x.fa"
    p.write_bytes(b">a desc here\r\nAC\r\nGT\r\n>\r\nTT\n>c\tz\nGG")
    d = sufr_amd.read_sequence_file(p)
    assert d.seq == b"ACGT%TT%GG$" and d.start_positions == [0, 5, 8] and d.sequence_names == ["a", "3", "c"]
    q = tmp_path / "x.fq"
    q.write_bytes(b"@r1 x\nACGT\n+\nIIII\n@r2\nTTGA\n+r2\nIIII\n")
    d = sufr_amd.read_sequence_file(q)
    assert d.seq == b"ACGT%TTGA$" and d.sequence_names == ["r1", "r2"]


def test_parallel_fasta_reader_equals_the_serial_one(tmp_path, monkeypatch):
    """Files of 32 MB and more are read by several threads (line-aligned slices, two sweeps): same text,
    start positions and names as the serial reader, with CRLF lines, blank lines, empty records, unnamed
    records and a last line without a newline."""
    rng = np.random.default_rng(12)
    acgt = np.frombuffer(b"ACGTNacgtn", dtype=np.uint8)
    parts = []
    for i in range(300):
        ln = int(rng.integers(0, 260_000))
        width = int(rng.choice([60, 70, 80, 1000]))
        eol = b"\r\n" if i % 7 == 3 else b"\n"
        hdr = b">" if i % 11 == 5 else b">seq%d some description" % i
        parts.append(hdr + eol)
        seq = acgt[rng.integers(0, acgt.size, size=ln)].tobytes()
        for o in range(0, ln, width):
            parts.append(seq[o:o + width] + eol)
        if i % 13 == 0:
            parts.append(eol)
    blob = b"".join(parts)
    blob = blob[:-1] if blob.endswith(b"\n") else blob
    assert len(blob) > (32 << 20)
    p = tmp_path / "big.fa"
    p.write_bytes(blob)
    par = sufr_amd.read_sequence_file(p)
    monkeypatch.setenv("SUFR_SERIAL_READER", "1")
    ser = sufr_amd.read_sequence_file(p)
    assert par.seq == ser.seq
    assert par.start_positions == ser.start_positions and par.sequence_names == ser.sequence_names
    assert len(par.sequence_names) == 300 and par.seq.endswith(b"$")


def test_compressed_input_reads_like_plain_text(tmp_path):
    """needletail (the reference's reader, util.rs:55) inflates gzip, bzip2 and xz transparently; so does this reader,
    concatenated members / streams included; damaged streams are refused with a message."""
    import gzip
    plain = (GOLDEN / "inputs" / "long_dna_sequence.fa").read_bytes()
    want = sufr_amd.read_sequence_file(GOLDEN / "inputs" / "long_dna_sequence.fa")
    one = tmp_path / "a.fa.gz"
    one.write_bytes(gzip.compress(plain))
    half = len(plain) // 2
    two = tmp_path / "b.fa.gz"
    two.write_bytes(gzip.compress(plain[:half]) + gzip.compress(plain[half:]))
    for p in (one, two):
        got = sufr_amd.read_sequence_file(p)
        assert (got.seq, got.start_positions, got.sequence_names) == (want.seq, want.start_positions,
                                                                        want.sequence_names)
    import bz2
    import lzma
    cases = {"c.fa.bz2": bz2.compress(plain), "d.fa.bz2": bz2.compress(plain[:half]) + bz2.compress(plain[half:]),
             "e.fa.xz": lzma.compress(plain), "f.fa.xz": lzma.compress(plain[:half]) + lzma.compress(plain[half:]),
             "g.fa.xz": lzma.compress(plain, preset=9 | lzma.PRESET_EXTREME)}
    for name, data in cases.items():
        (tmp_path / name).write_bytes(data)
        got = sufr_amd.read_sequence_file(tmp_path / name)
        assert (got.seq, got.start_positions, got.sequence_names) == (want.seq, want.start_positions, want.sequence_names), name
    big = b">r\n" + bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(1).integers(0, 4, 3_000_000)]) + b"\n"
    for name, data in (("h.fa.bz2", bz2.compress(big)), ("i.fa.xz", lzma.compress(big))):   # output far beyond the first buffer
        (tmp_path / name).write_bytes(data)
        assert sufr_amd.read_sequence_file(tmp_path / name).seq == big[3:-1] + b"$"
    for name, data in (("j.fa.bz2", bz2.compress(plain)[:-20]), ("k.fa.xz", lzma.compress(plain)[:-20]),
                       ("l.fa.bz2", b"BZh9" + b"\x00" * 40)):
        (tmp_path / name).write_bytes(data)
        with pytest.raises(sufr_amd.SufrHipError) as e:
            sufr_amd.read_sequence_file(tmp_path / name)
        assert "corrupt or truncated" in str(e.value), name


def test_empty_input_dies():  # cli.rs:103-110
    with pytest.raises(sufr_amd.SufrHipError):
        sufr_amd.read_sequence_file(GOLDEN / "inputs" / "empty.fa")
    with pytest.raises(sufr_amd.SufrHipError):
        sufr_amd.read_sequence_file(GOLDEN / "inputs" / "does_not_exist.fa")


@pytest.mark.parametrize("soft", [False, True])
def test_normalize_matches_oracle(oracle, soft):
    raw = np.random.default_rng(3).integers(0, 256, size=100_000, dtype=np.uint8)
    assert np.array_equal(sufr_amd.normalize(raw, soft), oracle.normalize(raw, soft))
    assert sufr_amd.normalize(b"ACGTacgtNn%$", False).tobytes() == b"ACGTACGTNN%$"
    assert sufr_amd.normalize(b"ACGTacgtNn%$", True).tobytes() == b"ACGTNNNNNN%$"


@pytest.mark.parametrize("name", sorted(GOLDEN_CASES))
def test_writer_reproduces_golden_bytes(tmp_path, name):
    """sufr_write_file fed with the golden file's own fields must give back the golden file."""
    g = parse_sufr(GOLDEN / "expected" / name)
    case = GOLDEN_CASES[name]
    out = tmp_path / name
    L = sufr_amd.lib()
    text = np.frombuffer(g.text, dtype=np.uint8)
    sa = np.ascontiguousarray(g.sa); lcp = np.ascontiguousarray(g.lcp)
    starts = np.asarray(g.sequence_starts, dtype=np.uint64)
    names = (C.c_char_p * len(g.sequence_names))(*[s.encode() for s in g.sequence_names])
    err = C.create_string_buffer(256)
    mask = case.get("seed_mask")
    rc = L.sufr_write_file(str(out).encode(), int(g.is_dna), int(g.allow_ambiguity), int(g.ignore_softmask),
                           text.ctypes.data, text.size, g.width, sa.ctypes.data, lcp.ctypes.data, sa.size,
                           0, 0, mask.encode() if mask else None, starts.ctypes.data, starts.size, names, err, 256)
    assert rc == 0, err.value
    assert out.read_bytes() == (GOLDEN / "expected" / name).read_bytes()


def test_writer_u64_width(tmp_path, oracle):
    """T = u64 layout (sequence_starts, SA, LCP 8 bytes wide) against the oracle's writer."""
    text = np.frombuffer(b"ACGTNNACGT$", dtype=np.uint8)
    sa = np.array([10, 6, 0, 7, 1, 8, 2, 5, 4, 9, 3], dtype=np.uint64)
    lcp = np.array([0, 0, 4, 0, 3, 0, 2, 0, 1, 0, 1], dtype=np.uint64)
    a, b = tmp_path / "a.sufr", tmp_path / "b.sufr"
    oracle.write_file(a, is_dna=True, allow_ambiguity=True, ignore_softmask=False, norm_text=text, sa=sa, lcp=lcp,
                      width=8, sequence_starts=[0], sequence_names=["1"])
    starts = np.array([0], dtype=np.uint64); names = (C.c_char_p * 1)(b"1"); err = C.create_string_buffer(256)
    rc = sufr_amd.lib().sufr_write_file(str(b).encode(), 1, 1, 0, text.ctypes.data, text.size, 8, sa.ctypes.data,
                                       lcp.ctypes.data, sa.size, 0, 0, None, starts.ctypes.data, 1, names, err, 256)
    assert rc == 0 and a.read_bytes() == b.read_bytes()


def test_writer_max_query_len_header(tmp_path):
    text = np.frombuffer(b"AC$", dtype=np.uint8)
    sa = np.array([2, 0, 1], dtype=np.uint32); lcp = np.zeros(3, dtype=np.uint32)
    starts = np.array([0], dtype=np.uint64); names = (C.c_char_p * 1)(b"s"); err = C.create_string_buffer(256)
    p = tmp_path / "m.sufr"
    rc = sufr_amd.lib().sufr_write_file(str(p).encode(), 1, 0, 0, text.ctypes.data, 3, 4, sa.ctypes.data,
                                       lcp.ctypes.data, 3, 1, 7, None, starts.ctypes.data, 1, names, err, 256)
    assert rc == 0
    g = parse_sufr(p)
    assert g.max_query_len == 7 and g.sa.tolist() == [2, 0, 1] and g.sequence_names == ["s"]


def test_lcp_pair():
    rng = np.random.default_rng(5)
    t = rng.integers(65, 67, size=5000, dtype=np.uint8)
    b = t.tobytes()
    for _ in range(300):
        i, j = (int(v) for v in rng.integers(0, t.size, size=2))
        if i == j:
            continue
        k = 0
        while i + k < len(b) and j + k < len(b) and b[i + k] == b[j + k]:
            k += 1
        assert sufr_amd.lcp_pair(t, i, j) == k


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_builder_rejects_max_query_len_with_seed_mask_even_when_zero():
    """SufrBuilder::new: `seed_mask.is_some() && max_query_len.is_some()` bails, Some(0) included
    (sufr_builder.rs:163-165)."""
    for mql in (0, 5):
        a = sufr_amd.SufrBuilderArgs(text=b"ACGTACGT$", max_query_len=mql, seed_mask="101", is_dna=True,
                                     sequence_starts=[0], sequence_names=["1"])
        with pytest.raises(sufr_amd.SufrHipError, match="Cannot use max_query_len and seed_mask together"):
            sufr_amd.SufrBuilder(a, write=False)


def test_build_fails_loudly_without_gpu(tmp_path):
    assert sufr_amd.lib().sufr_hip_device_count() == 0
    with pytest.raises(sufr_amd.SufrHipError) as e:
        sufr_amd.SufrBuilder(sufr_amd.SufrBuilderArgs(text=b"ACGT$", path=str(tmp_path / "o.sufr"), is_dna=True))
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    with pytest.raises(sufr_amd.SufrHipError):
        sufr_amd.create(str(GOLDEN / "inputs" / "1.fa"), str(tmp_path / "o.sufr"), is_dna=True)
    assert not (tmp_path / "o.sufr").exists()


def test_cli_usage_and_errors(tmp_path):
    exe = str(sufr_amd.CLI_PATH)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "Usage: sufr" in r.stderr
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--num-partitions" in r.stdout and "--ignore-softmask" in r.stdout
    r = subprocess.run([exe, "create", "-m", "3", "-s", "101", "x.fa"], capture_output=True, text=True)
    assert r.returncode == 2 and "cannot be used with" in r.stderr
    if not HAS_GPU:
        r = subprocess.run([exe, "create", "--dna", "-o", str(tmp_path / "o.sufr"), str(GOLDEN / "inputs" / "1.fa")],
                           capture_output=True, text=True)
        assert r.returncode == 1 and r.stderr.startswith("Error: ")


def test_no_flat_instructions_in_the_kernels():
    """Wave-private exchanges through LDS rely on ds_* ordering; a pointer that loses the LDS address space
    makes hipcc emit FLAT accesses, which are unordered.  The device ISA of every
    sufr kernel must be free of them."""
    csrc = sufr_amd.LIB_PATH.parents[1]
    r = subprocess.run(["make", "-C", str(csrc), "asm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = (csrc / "_build" / "sufr_kernels.s").read_text()
    bad = [ln for ln in asm.splitlines() if re.search(r"\bflat_(load|store|atomic)", ln)]
    assert not bad, bad[:5]
    # and no kernel spills to scratch
    assert all(int(v) == 0 for v in re.findall(r"\.private_segment_fixed_size:\s*(\d+)", asm))


def test_integration_md_names_every_export():
    """INTEGRATION.md is the binding a libsufr maintainer would write from: every function include/*.h declares is named there
    (the block of section 2, or the list of the exports it leaves out)"""
    from pathlib import Path
    ROOT = Path(__file__).resolve().parent.parent
    text = (ROOT / "INTEGRATION.md").read_text()
    missing = []
    for h in ("sufr_hip.h", "sufr_query.h"):
        src = (ROOT / "include" / h).read_text()
        for name in sorted(set(re.findall(r"\b(sufr_[a-z0-9_]+)\s*\(", src))):
            if f"}} {name};" in src or f"struct {name}" in src:       # (a type that a comment happens to follow with a bracket)
                continue
            if name not in text:
                missing.append(name)
    assert not missing, missing
