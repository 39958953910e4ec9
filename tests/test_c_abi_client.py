"""The public headers from plain C: tests/c_abi/abi_client.c includes include/sufr_hip.h and include/sufr_query.h,
links libsufr_hip.so and nothing else of this repository -- the position of a host in another language."""
import subprocess

import pytest

import sufr_amd
from oracle_helper import GOLDEN, REPO

EXP = GOLDEN / "expected"


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    out = tmp_path_factory.mktemp("abi") / "abi_client"
    lib_dir = sufr_amd.LIB_PATH.parent
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", str(REPO / "include"), str(REPO / "tests" / "c_abi" / "abi_client.c"),
                        "-o", str(out), "-L", str(lib_dir), "-lsufr_hip", f"-Wl,-rpath,{lib_dir}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_c_client_reads_and_searches(client):            # cli.rs:907-945 through C
    r = subprocess.run([str(client), "query", str(EXP / "2.sufr"), "AC", "GT", "XX"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["AC 1 5", "GT 9 13", "XX 0 0", "positions: 13 4 9 0", "names: ABC@0 DEF@9"]
    r = subprocess.run([str(client), "query", str(EXP / "nope.sufr"), "AC"], capture_output=True, text=True)
    assert r.returncode == 1 and "nope.sufr" in r.stderr


@pytest.mark.gpu
def test_c_client_builds_and_searches_on_the_device(client, tmp_path):
    r = subprocess.run([str(client), "device", str(EXP / "2.sufr"), "AC", "GT", "XX"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines()[:4] == ["AC 1 5", "GT 9 13", "XX 0 0", "positions: 13 4 9 0"]
    text = tmp_path / "t.txt"
    text.write_bytes(b"ACGTNNACGT$")                       # the text of 1.fa: the file must be 1.sufr
    out = tmp_path / "t.sufr"
    r = subprocess.run([str(client), "build", str(text), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("built 9 suffixes of 11 bytes")
    assert out.read_bytes() == (EXP / "1.sufr").read_bytes()
