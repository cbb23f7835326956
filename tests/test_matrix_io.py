"""The reference's matrix dump format (include/HPDDM_matrix.hpp:121-135, :173-244): reader and writer against files dumped
by the compiled reference (tests/golden/dump/out_{r}_4.txt: examples/schwarz.cpp -Nx 20 -Ny 20 on 4 ranks with
-hpddm_dump_matrices=out), and the P-row harness (examples/solver.py protocol) on them."""
import filecmp
import os
import subprocess
import sys

import numpy as np
import pytest

from hpddm_amd.generate import generate2d
from hpddm_amd.matrix_io import csrmv, read_matrix, write_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DUMP = os.path.join(HERE, "golden", "dump")


@pytest.mark.parametrize("rank", range(4))
def test_reader_matches_generator_and_writer_is_byte_identical(rank, tmp_path):
    path = os.path.join(DUMP, f"out_{rank}_4.txt")
    mat = read_matrix(path)
    assert (mat["n"], mat["m"], mat["sym"], mat["nnz"], mat["numbering"]) == (121, 121, False, 561, "C")
    sub = generate2d(20, 20, 4, overlap=1)[rank]  # bit-exact restatement of examples/generate.cpp
    assert np.array_equal(mat["ia"], sub["ia"]) and np.array_equal(mat["ja"], sub["ja"]) and np.array_equal(mat["a"], sub["a"])
    out = tmp_path / "rt.txt"
    write_matrix(out, mat["n"], mat["ia"], mat["ja"], mat["a"], sym=mat["sym"], numbering=mat["numbering"])
    assert filecmp.cmp(path, out, shallow=False)


def test_symmetric_storage_roundtrip(tmp_path):
    sub = generate2d(12, 12, 1, overlap=1, sym=True)[0]
    p = tmp_path / "s.txt"
    write_matrix(p, sub["n"], sub["ia"], sub["ja"], sub["a"], sym=True)
    mat = read_matrix(p)
    assert mat["sym"] and np.array_equal(mat["ja"], sub["ja"]) and np.array_equal(mat["a"], sub["a"])
    full = generate2d(12, 12, 1, overlap=1, sym=False)[0]
    x = np.random.default_rng(0).random(sub["n"])
    assert np.allclose(csrmv(mat, x), csrmv({**full, "sym": False}, x), rtol=1e-14)


def test_reader_accepts_every_variant_of_the_reference_parser(tmp_path):
    """include/HPDDM_matrix.hpp:173-244: one-, three- and four-field headers, '%' comments, blank lines, value-first entries,
    complex values"""
    ref = {"ia": [0, 2, 3], "ja": [0, 1, 1], "a": [4.0, -1.0, 3.0]}
    variants = {
        "five": "# c\n2 2 0  3 C\n1 1 4.0\n1 2 -1.0\n2 2 3.0\n",
        "four": "% c\n\n2 2 0 3\n1 1 4.0\n1 2 -1.0\n\n2 2 3.0\n",
        "three": "2 2 3\n% inside\n1 1 4.0\n1 2 -1.0\n2 2 3.0\n",
        "one": "2\n# then the number of entries\n3\n1 1 4.0\n1 2 -1.0\n2 2 3.0\n",
        "value_first": "2 2 0 3\n4.0 1 1\n-1.0 1 2\n3.0 2 2\n",
    }
    for name, text in variants.items():
        p = tmp_path / (name + ".txt")
        p.write_text(text)
        mat = read_matrix(p)
        assert (mat["n"], mat["m"], mat["sym"], mat["nnz"]) == (2, 2, False, 3), name
        assert list(mat["ia"]) == ref["ia"] and list(mat["ja"]) == ref["ja"] and list(mat["a"]) == ref["a"], name
    p = tmp_path / "z.txt"
    p.write_text("2 2 1 2\n1 1 (1.5,-2.0)\n2 2 (0.0,1.0)\n")
    mat = read_matrix(p)
    assert mat["sym"] and mat["a"].dtype == np.complex128 and mat["a"][0] == 1.5 - 2.0j and mat["a"][1] == 1.0j


def test_reader_rejects_malformed(tmp_path):
    p = tmp_path / "bad.txt"
    p.write_text("# c\n# c\n2 2 0  2 C\n1 1 1.0\n")
    with pytest.raises(ValueError):
        read_matrix(p)
    p.write_text("# c\n# c\n2 2 0  1 C\n3 1 1.0\n")
    with pytest.raises(ValueError):
        read_matrix(p)


@pytest.mark.gpu
@pytest.mark.parametrize("rank", [0, 3])
def test_solver_py_protocol(rank):
    """examples/solver.py:32-50: numfact + solve of a random right-hand side, residual <= 1e-8, exit status 0"""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "solver.py"), os.path.join(DUMP, f"out_{rank}_4.txt")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "--- residual" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_local_solver_benchmark_protocol():
    """benchmark/local_solver.cpp: one row per trial, one column per nu"""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "local_solver.py"), os.path.join(DUMP, "out_1_4.txt"), "--rhs", "4", "--solve-phase-only"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    rows = [ln for ln in res.stdout.splitlines() if "|" in ln]
    assert len(rows) == 3 and all(len(r.split("|")[0].split()) == 3 for r in rows), res.stdout


def test_dump_matrices_option_writes_the_reference_files(tmp_path):
    """-hpddm_dump_matrices=<prefix> (include/HPDDM_subdomain.hpp:370-388): destroying the operator leaves one file per subdomain,
    byte for byte what the compiled reference wrote for the same problem (tests/golden/dump).  Host code only."""
    from hpddm_amd import hpddm
    subs = generate2d(20, 20, 4, overlap=1)
    prefix = str(tmp_path / "out")
    A, d = hpddm.schwarz_from_subdomains(subs, options=f"-hpddm_dump_matrices={prefix} -hpddm_tol 1e-8")
    assert A.get_option("tol") == 1e-8
    A.destroy()
    for r in range(4):
        with open(f"{prefix}_{r}_4.txt", "rb") as new, open(os.path.join(DUMP, f"out_{r}_4.txt"), "rb") as ref:
            assert new.read() == ref.read()


@pytest.mark.gpu
def test_schwarz_py_example_reproduces_the_reference_numbers():
    """examples/schwarz.py, the counterpart of the reference's own Python example: 19 iterations and the residual the reference
    prints for -Nx 40 -Ny 40 on 4 subdomains (tests/golden/p40_onelevel), exit status 0; Block GMRES on two random right-hand sides
    with the deflated coarse correction; the single-subdomain direct solve"""
    ex = os.path.join(ROOT, "examples", "schwarz.py")
    res = subprocess.run([sys.executable, ex, "--subdomains", "4", "-Nx", "40", "-Ny", "40", "-hpddm_verbosity=1"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "--- residual = 1.428088e-05 / 3.998092e+01" in res.stdout and "GMRES converges after 19 iterations" in res.stdout
    res = subprocess.run([sys.executable, ex, "--subdomains", "4", "-Nx", "40", "-Ny", "40", "-hpddm_schwarz_coarse_correction", "deflated", "-hpddm_geneo_nu=0",
                          "-generate_random_rhs", "2", "-hpddm_krylov_method", "bgmres", "-hpddm_verbosity=1"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "(rhs #2)" in res.stdout and "BGMRES converges" in res.stdout, res.stdout + res.stderr
    res = subprocess.run([sys.executable, ex, "--subdomains", "1", "-Nx", "30", "-Ny", "30"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "--- residual" in res.stdout, res.stdout + res.stderr


def test_schwarz_py_example_fails_loudly_without_a_gpu():
    """no CPU fallback behind the example either: without a device the first device call raises"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ex = os.path.join(ROOT, "examples", "schwarz.py")
    res = subprocess.run([sys.executable, ex, "--subdomains", "4", "-Nx", "20", "-Ny", "20"], capture_output=True, text=True, timeout=600)
    assert res.returncode != 0 and "no ROCm-capable device" in res.stderr
