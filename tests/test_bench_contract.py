"""The bench contract on the committed line (profiles/r05_bench_default_stdout.json, printed by `python bench.py` on an MI355X):
every key the driver reads is there, the workload is the one BASELINE.json quotes its target on (configs[2], real GenEO space) and
the derived fields are consistent with each other.  CPU only."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r05_bench_default_stdout.json")) as fh:
        rows = [ln for ln in fh if ln.startswith('{"metric"')]
    assert len(rows) == 1, "bench.py prints ONE JSON line"
    return json.loads(rows[0])


def test_contract_keys_and_consistency():
    d = _line()
    with open(os.path.join(ROOT, "BASELINE.json")) as fh:
        base = json.load(fh)
    assert base["metric"].startswith("RAS-precond applies/sec")          # the headline metric of BASELINE.json ...
    assert d["metric"] == "ras_precond_applies_per_sec" and d["unit"] == "applies/s"   # ... under the name bench.py prints it
    for key in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["workload"].startswith("BASELINE.json configs[2]: 3-D Poisson 256^3") and "GenEO" in d["config"]["workload"]
    assert d["two_level"]["coarse_space"].startswith("GenEO") and d["two_level"]["coarse_dim"] == 160
    assert d["one_level"]["gmres"]["iterations"] == 38 and d["two_level"]["gmres"]["iterations"] == 20
    assert d["configs_1"]["workload"].startswith("BASELINE.json configs[1]") and d["configs_1"]["gmres"]["iterations"] == 26
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]          # one apply per step
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["bytes_alg_per_sweep"] / r["seconds_per_sweep"] / 1e9) <= 1e-9 * r["achieved"]
    n, nnz = d["config"]["n_dof_per_gpu"], d["config"]["nnz_L_per_gpu"]
    assert r["bytes_alg_per_sweep"] == 2.0 * nnz * 8.0 + 4.0 * n * 8.0            # 2 nnz(L) sizeof(K) + 4 n mu sizeof(K), mu = 1
    assert r["traffic"] is None or r["traffic"] >= r["bytes_alg_per_sweep"]         # measured HBM bytes cannot be below the algorithmic ones
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == d["unit"] and c["value"] > 0 and "sample" in c


def test_traffic_profile_matches_the_line():
    d = _line()
    with open(os.path.join(ROOT, "profiles", "r05_pmc_traffic_c3.json")) as fh:
        t = json.load(fh)
    assert t["traffic_bytes"] == (2 * t["FETCH_SIZE_KB_per_sweep"] + t["WRITE_SIZE_KB_per_sweep"]) * 1024
    # the line quotes the traffic file of this round's PMC passes (scripts/r05_profiles.sh pmc), collected on the same build just before it ran
    assert d["roofline"]["traffic_source"].startswith("profiles/r05_pmc_traffic_c3.json") and d["roofline"]["traffic_measured_in_this_run"] is False
    assert d["roofline"]["traffic"] == t["traffic_bytes"] and t["algorithmic_bytes"] == d["roofline"]["bytes_alg_per_sweep"]
    assert 1.0 <= t["traffic_over_algorithmic"] <= 1.15


def test_round_4_keys():
    """what round 4 added to the line: the host-pointer boundary beside (never in) `value`, the sampled CPU baseline, configs[4]'s share on
    the Helmholtz problem SURVEY 8(d) C5 defines (absorbing boundary, ORAS, DtN coarse space from the complex solveGEVP)"""
    d = _line()
    h = d["host_pointer_boundary"]
    assert h["apply_ms"] > d["ms_per_step"] and abs(h["applies_per_sec"] - 1e3 / h["apply_ms"]) <= 1e-6 * h["applies_per_sec"]
    assert h["bytes_over_pcie"] == 2.0 * 8.0 * d["config"]["n_dof_per_gpu"] and 20.0 < h["effective_GBps"] < 64.0      # PCIe gen 5 x16
    c = d["cpu_baseline"]
    assert c["sample_subdomains"] == 2 and abs(c["sample_scale"] - 4.0) < 0.05 and "2 of its 8 subdomains" in c["sample"]
    assert d["config"]["setup_seconds_of_which_plain_factor_for_cpu_baseline"] == 0.0
    s4 = d["configs_4_share"]
    assert s4["dtype"] == "c128" and "absorbing boundary" in s4["workload"] and "ORAS" in s4["workload"] and "DtN" in s4["workload"]
    assert s4["two_level"]["coarse_space"].startswith("DtN: solveGEVP") and s4["two_level"]["gmres"]["method"] == "bgmres" and s4["two_level"]["gmres"]["rhs"] == 8
    assert 0 < s4["two_level"]["gmres"]["iterations"] < 60
    m8 = d["two_level"]["deflation_mfma_mu8"]
    assert m8["panel_GBps"] > 3500.0 and "r04_pmc_mfma_deflation" in m8["counters"]


def test_round_5_keys():
    """what round 5 added: a CPU leg for the complex share (complex substitutions of the port, one thread per subdomain), `cores` = the threads
    `value` was measured on (said in `cores_used_by`), the deflation of the complex share on the direct MFMA kernels"""
    d = _line()
    c = d["cpu_baseline"]
    assert c["cores"] == 8 and "cores_used_by" in c and "usable_cpus" in c["cores_used_by"]
    s4 = d["configs_4_share"]
    cz = s4["cpu_baseline"]
    assert cz["kind"] == "port" and cz["unit"] == "applies/s" and 0 < cz["value"] < s4["value"] and cz["cores"] == 8 and "complex substitutions" in cz["sample"]
    assert s4["two_level"]["deflation_ms"] < 0.2 and abs(s4["two_level"]["gmres"]["iterations"] - 20) <= 1
    assert d["roofline"]["frac"] > 0.70 and d["config"]["launches_per_sptrsv"] == 176.0


def test_dry_run_prints_the_layout_of_the_multi_gpu_configs_without_a_gpu():
    """`bench.py --gpus N --dry-run`: every rank's brick of subdomains, the partition handed to the library and the peer GPUs its halo
    lists name -- configs[3]'s layout (8 GPUs: every GPU a neighbour of the 7 others, one peer per xGMI link) and configs[4]'s (4 GPUs)
    at a small size, on CPU"""
    import subprocess
    import sys
    for extra, peers in ((["--gpus", "8", "--problem", "elasticity", "--grid", "8"], {"7": 8}), (["--gpus", "4", "--problem", "helmholtz", "--grid", "8", "--mu", "8"], {"3": 4})):
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"] + extra, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-400:]
        d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["dry_run"] is True and d["peer_gpus_histogram"] == peers and d["subdomains"] == 8 * d["n_gpus"]
        assert all(r not in p for r, p in enumerate(d["peer_gpus_of_rank"])) and len(set(d["n_dof_per_gpu"])) == 1
