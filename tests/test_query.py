"""The reader / query side, host part: no GPU.

* the `sufr` binary's count / locate / extract / list / summarize print what the reference's CLI tests expect
  (sufr/tests/cli.rs:305-1164; the .out files are the reference's data/expected/*.out, copied as data);
* the Python mirror returns the reference's result structures (sufr_file.rs:1186-1500, suffix_array.rs doc tests);
* sufr_file_search agrees with a brute-force witness that scans every rank, on the golden files and on files the oracle
  writes (max_query_len builds and seed-mask builds included);
* malformed files are refused with a message.
"""
import re
import subprocess
import zlib

import numpy as np
import pytest

import sufr_amd
from sufr_amd import SufrFile, SuffixArray
from sufr_amd.sufr_file import ExtractSequence, LocatePosition
from oracle_helper import GOLDEN, parse_sufr

EXP = GOLDEN / "expected"
SUFR1, SUFR2, SUFR3 = EXP / "1.sufr", EXP / "2.sufr", EXP / "3.sufr"
UNIPROT, UNIPROT_MASKED, LONG = EXP / "uniprot.sufr", EXP / "uniprot-masked.sufr", EXP / "long_dna_sequence.sufr"


def run(*args, check=True):
    r = subprocess.run([str(sufr_amd.CLI_PATH), *map(str, args)], capture_output=True, text=True)
    if check:
        assert r.returncode == 0, r.stderr
    return r


def test_query_header_symbols_are_exported():
    hdr = (sufr_amd.LIB_PATH.parents[3] / "include" / "sufr_query.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sufr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(sufr_amd.QUERY_EXPORTS), declared ^ set(sufr_amd.QUERY_EXPORTS)
    nm = subprocess.run(["nm", "-D", "--defined-only", str(sufr_amd.LIB_PATH)], capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(rf"\bT {name}\b", nm), name


# ---------------------------------------------------------------------------------------------------------------------
# CLI: the reference's expectations, each with the memory switches its tests loop over
# ---------------------------------------------------------------------------------------------------------------------
MEM3 = [[], ["-l"], ["-v"]]


@pytest.mark.parametrize("mem", MEM3)
def test_cli_count(mem):                     # cli.rs:339-360
    assert run("count", SUFR1, *mem, "AC", "X", "GT").stdout == "AC 2\nX 0\nGT 2\n"
    qs = ["AAAAAAA", "TGTCTC", "TGATAGCAGCTTCTGAACTGGTTACCTGCCGTGAGT"]
    assert run("co", SUFR3, *mem, *qs).stdout == "".join(f"{q} 1\n" for q in qs)


EXTRACT_CASES = [  # cli.rs:412-650
    (SUFR1, ["AC", "GT", "XX"], [], [">1:6-11 AC 0", "ACGT$", ">1:0-11 AC 0", "ACGTNNACGT$", ">1:8-11 GT 0", "GT$",
                                      ">1:2-11 GT 0", "GTNNACGT$"], "XX not found\n"),
    (SUFR1, ["AC", "GT"], ["-p", 1], [">1:5-11 AC 1", "NACGT$", ">1:0-11 AC 0", "ACGTNNACGT$", ">1:7-11 GT 1", "CGT$",
                                       ">1:1-11 GT 1", "CGTNNACGT$"], ""),
    (SUFR1, ["AC", "GT"], ["-s", 3], [">1:6-9 AC 0", "ACG", ">1:0-3 AC 0", "ACG", ">1:8-11 GT 0", "GT$", ">1:2-5 GT 0", "GTN"], ""),
    (SUFR1, ["AC", "GT"], ["-p", 1, "-s", 3], [">1:5-9 AC 1", "NACG", ">1:0-3 AC 0", "ACG", ">1:7-11 GT 1", "CGT$",
                                                 ">1:1-5 GT 1", "CGTN"], ""),
    (UNIPROT, ["RNELNNEEA", "DTPTNCPT", "GSGLSLLSD"], ["-p", 4, "-s", 12],
     [">sp|Q9U408|14331_ECHGR:38-54 RNELNNEEA 4", "MAKMRNELNNEEANLL", ">sp|Q6GZX3|002L_FRG3G:218-234 DTPTNCPT 4", "GTQRDTPTNCPTQVCQ",
      ">sp|Q6GZW6|009L_FRG3G:390-406 GSGLSLLSD 4", "EYVNGSGLSLLSDILL"], ""),
    (UNIPROT, ["RNELNNEEA"], ["-s", 10, "-m", 3],
     [">sp|Q6GZW1|014R_FRG3G:111-121 RNELNNEEA 0", "RNEEDDDG%M", ">sp|P0C9G4|1101L_ASFP4:80-90 RNELNNEEA 0", "RNEFCTYYVT",
      ">sp|P0C9G1|1101L_ASFWA:80-90 RNELNNEEA 0", "RNEFCTYYVT", ">sp|Q9U408|14331_ECHGR:42-52 RNELNNEEA 0", "RNELNNEEAN",
      ">sp|O55726|110R_IIV6:31-41 RNELNNEEA 0", "RNEPSHYQTV", ">sp|Q196U3|117L_IIV3:27-37 RNELNNEEA 0", "RNEYDNAVAS",
      ">sp|Q197E9|011L_IIV3:55-65 RNELNNEEA 0", "RNEYNKVHIE"], ""),
    (UNIPROT_MASKED, ["RNELNNEEA"], ["-s", 10, "-m", 3],
     [">sp|P32234|128UP_DROME:38-48 RNELNNEEA 0", "RRELISPKGG", ">sp|P26709|1107L_ASFL5:128-138 RNELNNEEA 0", "RKELKKDEF%",
      ">sp|P0DO85|10H_STRNX:151-161 RNELNNEEA 0", "RLELLKHIRV", ">sp|Q9U408|14331_ECHGR:42-52 RNELNNEEA 0", "RNELNNEEAN",
      ">sp|P19084|11S3_HELAN:355-365 RNELNNEEA 0", "RGELRPNAIQ", ">sp|Q6GZV8|017L_FRG3G:18-28 RNELNNEEA 0", "RGELSALSAA",
      ">sp|Q6GZW6|009L_FRG3G:653-663 RNELNNEEA 0", "RLELSAPYGS"], ""),
]


@pytest.mark.parametrize("mem", MEM3)
@pytest.mark.parametrize("case", range(len(EXTRACT_CASES)))
def test_cli_extract(case, mem):
    path, queries, opts, lines, err = EXTRACT_CASES[case]
    r = run("extract", path, *mem, *opts, *queries)
    assert r.stdout == "".join(l + "\n" for l in lines)
    assert r.stderr == err


SUFFIXES1 = ["$", "ACGT$", "ACGTNNACGT$", "CGT$", "CGTNNACGT$", "GT$", "GTNNACGT$", "T$", "TNNACGT$"]
SA1, LCP1 = [10, 6, 0, 7, 1, 8, 2, 9, 3], [0, 0, 4, 0, 3, 0, 2, 0, 1]
LIST_CASES = [  # cli.rs:696-870
    ([], SUFFIXES1),
    (["-s"], [f"{s:>2} {x}" for s, x in zip(SA1, SUFFIXES1)]),
    (["-p"], [f"{l:>2} {x}" for l, x in zip(LCP1, SUFFIXES1)]),
    (["-s", "-r", "-p"], [f"{r:>2} {s:>2} {l:>2} {x}" for r, (s, l, x) in enumerate(zip(SA1, LCP1, SUFFIXES1))]),
    (["-rsp"], [f"{r:>2} {s:>2} {l:>2} {x}" for r, (s, l, x) in enumerate(zip(SA1, LCP1, SUFFIXES1))]),
    (["-r"], [f"{r:>2} {x}" for r, x in enumerate(SUFFIXES1)]),
    (["--len", 3], ["$", "ACG", "ACG", "CGT", "CGT", "GT$", "GTN", "T$", "TNN"]),
    (["-n", 3], ["$", "ACGT$", "ACGTNNACGT$"]),
]


@pytest.mark.parametrize("mem", [[], ["-v"]])
@pytest.mark.parametrize("case", range(len(LIST_CASES)))
def test_cli_list(case, mem):
    opts, lines = LIST_CASES[case]
    assert run("list", SUFR1, *mem, *opts).stdout == "".join(l + "\n" for l in lines)


def test_cli_list_ranks(tmp_path):          # parse_pos / parse_index, lib.rs:556-590
    assert run("ls", SUFR1, "1", "3-5", "0,8").stdout == "".join(SUFFIXES1[i] + "\n" for i in (1, 3, 4, 5, 0, 8))
    r = run("ls", SUFR1, "0", "20")
    assert r.stdout == "$\n" and r.stderr == "Invalid rank: 20\n"
    r = run("ls", SUFR1, "5-2", check=False)
    assert r.returncode == 1 and "First number in range (5) must be lower than second number (2)" in r.stderr
    r = run("ls", SUFR1, "x", check=False)
    assert r.returncode == 1 and 'illegal list value: "x"' in r.stderr
    out = tmp_path / "ls.txt"
    assert run("ls", SUFR1, "-o", out, "-n", 2).stdout == "" and out.read_text() == "$\nACGT$\n"


LOCATE_CASES = [  # cli.rs:907-1120
    (SUFR2, ["AC", "GT"], [], "locate1.out"),
    (SUFR2, ["AC", "GT"], ["-a"], "locate-abs.out"),
    (UNIPROT, ["RNELNNEEA", "DTPTNCPT", "GSGLSLLSD"], [], "uniprot-search1.out"),
    (UNIPROT_MASKED, ["RNEL", "DTPT", "GSGL"], [], "uniprot-search-masked.out"),
    (UNIPROT_MASKED, ["RNELNNEEA", "DTPTNCPT", "GSGLSLLSD"], ["-m", 3], "uniprot-search-masked-mql-3.out"),
    (UNIPROT_MASKED, ["RNEL"], ["-a"], "uniprot-search-masked-absolute.out"),
    (LONG, ["CATGTTGTCACG", "CCATGGGAC", "GGATGAAGAAAAGCA"], [], "locate_long_dna.out"),
    (LONG, ["CATGTTGTCACG", "CCATGGGAC", "GGATGAAGAAAAGCA"], ["-m", 6], "locate_long_dna_mql_6.out"),
]


@pytest.mark.parametrize("mem", MEM3)
@pytest.mark.parametrize("case", range(len(LOCATE_CASES)))
def test_cli_locate(case, mem):
    path, queries, opts, expected = LOCATE_CASES[case]
    assert run("locate", path, *opts, *mem, *queries).stdout == (EXP / expected).read_text()


def test_cli_queries_from_a_file(tmp_path):      # parse_locate_queries, lib.rs:449-466
    q = tmp_path / "queries.txt"
    q.write_text("AC GT\nX\n")
    assert run("count", SUFR1, q).stdout == "AC 2\nGT 2\nX 0\n"
    r = run("lo", SUFR2, "-a", q)
    assert r.stdout == (EXP / "locate-abs.out").read_text() and r.stderr == "X not found\n"


def test_cli_summarize():                    # cli.rs:1126-1164
    out = run("summarize", SUFR1).stdout
    want = {"File Size": "172 bytes", "File Version": "6", "DNA": "true", "Allow Ambiguity": "false", "Ignore Softmask": "false",
            "Text Length": "11", "Len Suffixes": "9", "Max query len": "0", "Num sequences": "1", "Sequence starts": "0",
            "Sequence names": "1"}
    for row, value in want.items():
        m = re.search(rf"[|] {row}\s+[|] ([^|]+)", out)
        assert m and m.group(1).strip() == value, (row, out)
    masked = run("su", UNIPROT_MASKED).stdout
    m = re.search(r"[|] Seed mask\s+[|] ([^|]+)", masked)
    assert m and m.group(1).strip() == SufrFile(UNIPROT_MASKED).seed_mask
    assert "Max query len" not in masked
    # every line of the table has the same width, long cells are wrapped at 40 columns
    widths = {len(l) for l in masked.splitlines()}
    assert len(widths) == 1
    names = re.findall(r"^[|] (?:Sequence names)?\s+[|] (.+?)\s*[|]$", masked, flags=re.M)
    assert all(len(n) <= 40 or " " not in n for n in names)


def test_cli_errors():
    r = run("count", EXP / "nope.sufr", "AC", check=False)
    assert r.returncode == 1 and r.stderr.startswith("Error: ") and "nope.sufr" in r.stderr
    assert run("count", SUFR1, check=False).returncode == 2
    assert run("frobnicate", SUFR1, check=False).returncode == 2


# ---------------------------------------------------------------------------------------------------------------------
# library mirror: the reference's result structures
# ---------------------------------------------------------------------------------------------------------------------
def test_extract_result_structures():        # sufr_file.rs:1186-1256
    f = SufrFile(SUFR1)
    res = f.extract(["AC", "GT", "XX"], prefix_len=1, suffix_len=3)
    assert [(r.query_num, r.query) for r in res] == [(0, "AC"), (1, "GT"), (2, "XX")]
    assert res[0].sequences == [ExtractSequence(6, 1, "1", 0, (5, 9), 1), ExtractSequence(0, 2, "1", 0, (0, 3), 0)]
    assert res[1].sequences == [ExtractSequence(8, 5, "1", 0, (7, 11), 1), ExtractSequence(2, 6, "1", 0, (1, 5), 1)]
    assert res[2].sequences == []
    # suffix_array.rs doc test of extract
    res = SuffixArray.read(str(SUFR1), True).extract(["CGT", "GG"], prefix_len=1)
    assert res[0].sequences == [ExtractSequence(7, 3, "1", 0, (6, 11), 1), ExtractSequence(1, 4, "1", 0, (0, 11), 1)]
    assert res[1].sequences == []


def test_locate_result_structures():         # sufr_file.rs:1260-1500
    f = SufrFile(EXP / "abba.sufr")
    suf_by_rank = [14, 0, 12, 10, 1, 3, 5, 7, 13, 11, 9, 2, 4, 6, 8]
    assert f.suffix_array.tolist() == suf_by_rank          # test_file_access, 1505-1560
    for low_memory in (True, False):
        for q, ranks in (("A", range(1, 8)), ("B", range(8, 15)), ("ABAB", range(3, 7)), ("ABABB", [6]), ("BBBB", [])):
            (res,) = f.locate([q], low_memory=low_memory)
            assert res.query == q and res.query_num == 0
            assert res.positions == [LocatePosition(suf_by_rank[r], r, "1", suf_by_rank[r]) for r in ranks]


def test_suffix_array_facade_doc_examples():  # suffix_array.rs:140-440
    sa = SuffixArray.read(str(SUFR1), True)
    assert [(c.query, c.count) for c in sa.count(["AC", "GG", "CG"])] == [("AC", 2), ("GG", 0), ("CG", 2)]
    (hit, miss) = sa.locate(["ACG", "GGC"])
    assert hit.positions == [LocatePosition(6, 1, "1", 6), LocatePosition(0, 2, "1", 0)] and miss.positions == []
    assert sa.list(show_rank=True, show_suffix=True, show_lcp=True) == [
        " 0 10  0 $", " 1  6  0 ACGT$", " 2  0  4 ACGTNNACGT$", " 3  7  0 CGT$", " 4  1  3 CGTNNACGT$", " 5  8  0 GT$",
        " 6  2  2 GTNNACGT$", " 7  9  0 T$", " 8  3  1 TNNACGT$"]
    m = sa.metadata()
    assert (m.file_size, m.file_version, m.is_dna, m.allow_ambiguity, m.ignore_softmask, m.text_len, m.len_suffixes,
            m.num_sequences, m.sequence_starts, m.sequence_names, m.max_query_len, m.seed_mask) == \
        (172, 6, True, False, False, 11, 9, 1, [0], ["1"], 0, None)
    assert sa.string_at(0) == "ACGTNNACGT$" and sa.string_at(6, 3) == "ACG"


@pytest.mark.parametrize("name", sorted(p.name for p in EXP.glob("*.sufr")))
def test_reader_agrees_with_the_test_side_decoder(name):
    want = parse_sufr(EXP / name)
    f = SufrFile(EXP / name)
    assert (f.text_len, f.len_suffixes, f.index_width, f.is_dna, f.allow_ambiguity, f.ignore_softmask, f.max_query_len) == \
        (want.text_len, want.num_suffixes, want.width, want.is_dna, want.allow_ambiguity, want.ignore_softmask, want.max_query_len)
    assert f.sequence_starts == want.sequence_starts and f.sequence_names == want.sequence_names
    assert bytes(f.text) == want.text and np.array_equal(f.suffix_array, want.sa) and np.array_equal(f.lcp, want.lcp)
    assert (f.seed_mask or "") == "".join("1" if b else "0" for b in want.seed_mask)


# ---------------------------------------------------------------------------------------------------------------------
# search against a brute-force witness
# ---------------------------------------------------------------------------------------------------------------------
def witness_ranks(f: SufrFile, q: bytes, mql):
    """Ranks whose suffix matches `q`, by scanning every rank.  Plain files: the first min(len(q), L) symbols agree,
    L the effective max_query_len.  Seed-mask files: every care position below len(q) (the first L of them) agrees."""
    text, sa, n = bytes(f.text), f.suffix_array, f.text_len
    out = []
    if f.seed_mask is None:
        built = f.max_query_len
        eff = min(built, mql) if (built > 0 and mql is not None) else (mql if mql is not None else built)
        need = min(len(q), eff) if eff > 0 else len(q)
        for r, s in enumerate(sa.tolist()):
            if text[s:s + need] == q[:need] and s + need <= n:
                out.append(r)
    else:
        care = [i for i, c in enumerate(f.seed_mask) if c == "1"]
        if mql:
            care = care[:mql]
        care = [p for p in care if p < len(q)]
        for r, s in enumerate(sa.tolist()):
            if all(s + p < n and text[s + p] == q[p] for p in care):
                out.append(r)
    return out


def random_queries(rng, f: SufrFile, count, max_len):
    text = bytes(f.text)
    alphabet = sorted(set(text))
    qs = []
    for _ in range(count):
        kind = rng.integers(0, 4)
        L = int(rng.integers(1, max_len + 1))
        if kind < 2:                                   # a substring of the text
            at = int(rng.integers(0, len(text)))
            q = bytearray(text[at:at + L])
            if kind == 1 and q:                        # ... with one symbol changed
                q[int(rng.integers(0, len(q)))] = alphabet[int(rng.integers(0, len(alphabet)))]
            qs.append(bytes(q))
        else:
            qs.append(bytes(alphabet[int(i)] for i in rng.integers(0, len(alphabet), L)))
    return [q for q in qs if q]


def check_against_witness(f: SufrFile, queries, mql):
    hits = 0
    for q in queries:
        want = witness_ranks(f, q, mql)
        got = f.search(q, mql)
        if not want:
            assert got is None, (q, mql, got)
        else:
            assert want == list(range(want[0], want[-1] + 1)), (q, mql)      # the file is sorted for this comparison
            assert got == (want[0], want[-1] + 1), (q, mql, got, want[0], want[-1] + 1)
            hits += 1
    return hits


@pytest.mark.parametrize("name", ["1.sufr", "2.sufr", "2d.sufr", "2n.sufr", "2ns.sufr", "3.sufr", "abba.sufr", "long_dna_sequence.sufr",
                                  "long_dna_sequence_allow_ambiguity.sufr"])
@pytest.mark.parametrize("mql", [None, 1, 3, 7])
def test_search_equals_witness_on_golden_files(name, mql):
    f = SufrFile(EXP / name)
    rng = np.random.default_rng(zlib.crc32(f"{name}{mql}".encode()))
    queries = random_queries(rng, f, 40 if f.text_len > 2000 else 120, 12)
    assert check_against_witness(f, queries, mql) > 0


@pytest.mark.parametrize("mql", [None, 2, 3])
def test_search_equals_witness_on_the_masked_golden_file(mql):
    f = SufrFile(UNIPROT_MASKED)
    rng = np.random.default_rng(5)
    queries = random_queries(rng, f, 25, len(f.seed_mask))          # queries longer than the mask: see DESIGN.md 10
    assert check_against_witness(f, queries, mql) > 0


@pytest.mark.parametrize("build", [dict(max_query_len=4), dict(max_query_len=9), dict(seed_mask="1101"), dict(seed_mask="110011"),
                                   dict(seed_mask="10001")])
@pytest.mark.parametrize("mql", [None, 2, 6])
def test_search_equals_witness_on_truncated_and_masked_builds(oracle, tmp_path, build, mql):
    rng = np.random.default_rng(11)
    seq = "".join("ACGT"[i] for i in rng.integers(0, 4, 1500))
    seq = seq[:700] + seq[100:400] + seq[700:]                      # a long repeat: ties inside the truncated order
    fa = tmp_path / "x.fa"
    fa.write_text(">a\n" + seq[:1000] + "\n>b\n" + seq[1000:] + "\n")
    out = tmp_path / "x.sufr"
    oracle.create(fa, out, is_dna=True, **build)
    f = SufrFile(out)
    assert f.max_query_len == build.get("max_query_len", 0) and f.seed_mask == build.get("seed_mask")
    max_len = len(build["seed_mask"]) if "seed_mask" in build else 14
    assert check_against_witness(f, random_queries(rng, f, 60, max_len), mql) > 0


@pytest.mark.parametrize("threads", [1, 3, 0])
def test_threaded_batch_equals_single_searches(threads):
    f = SufrFile(LONG)
    rng = np.random.default_rng(8)
    queries = random_queries(rng, f, 3000, 14)
    for mql in (None, 5):
        lo, hi = f.search_batch(queries, mql, threads=threads)
        for i, q in enumerate(queries):
            r = f.search(q, mql)
            assert (int(lo[i]), int(hi[i])) == (r if r else (0, 0))
    lo, hi = f.search_batch([], threads=threads)
    assert lo.size == 0
    assert run("-t", 2, "count", SUFR1, "AC", "X", "GT").stdout == "AC 2\nX 0\nGT 2\n"


def test_empty_query_matches_every_suffix():
    f = SufrFile(SUFR1)
    assert f.search(b"") == (0, 9)


# ---------------------------------------------------------------------------------------------------------------------
# malformed files
# ---------------------------------------------------------------------------------------------------------------------
def test_malformed_files_are_refused(tmp_path):
    good = SUFR2.read_bytes()

    def opens(b):
        p = tmp_path / "bad.sufr"
        p.write_bytes(b)
        try:
            SufrFile(p).close()
            return None
        except sufr_amd.SufrHipError as e:
            return e.message

    assert opens(good) is None
    assert "too short" in opens(good[:40])
    assert "version 5" in opens(bytes([5]) + good[1:])
    assert "corrupt" in opens(good[:-3])                                   # sequence names cut
    assert "corrupt" in opens(good[:100])                                  # sections past the end
    b = bytearray(good); b[52:60] = (2**40).to_bytes(8, "little")          # absurd num_sequences
    assert "corrupt" in opens(bytes(b))
    b = bytearray(good); b[36:44] = (2**50).to_bytes(8, "little")          # absurd len_suffixes
    assert "corrupt" in opens(bytes(b))
    with pytest.raises(sufr_amd.SufrHipError):
        SufrFile(tmp_path / "missing.sufr")


def test_views_pin_the_mapping_and_close_is_a_request_until_they_are_gone():
    """SufrFile views alias the mapped file: close() with a view alive is deferred (pending_close), the file is unmapped
    when the last view goes (closed); a copy does not pin anything; a closed file hands out no more views."""
    import gc
    with SufrFile(str(EXP / "1.sufr")) as f:
        sa = f.suffix_array
        kept = f.array("lcp", copy=True)
    assert f.pending_close and not f.closed
    assert sa.tolist() == parse_sufr(EXP / "1.sufr").sa.tolist()      # still readable: the mapping is alive
    del sa
    gc.collect()
    assert f.closed and not f.pending_close
    assert kept.tolist() == parse_sufr(EXP / "1.sufr").lcp.tolist()
    with pytest.raises(ValueError):
        f.suffix_array
    g = SufrFile(str(EXP / "1.sufr"))
    g.close()
    assert g.closed
