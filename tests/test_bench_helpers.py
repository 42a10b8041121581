"""CPU: the host-side logic of bench.py that needs no GPU -- the parser of the host CLI's output (`configs` legs), the sweep-traffic
guard, the CPU topology record and the CPU sptrsv leg on a small grid (the reference's serial native_sptrsv from oracle/_ref where
it is present, else the oracle's restatement: kernels.hpp:54-107)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CLI_STDOUT = """
               Residual Norms                           Time for iteration
+------------------------------------------+        +-------------------------+
||A*x_0 - b||_2 = 1.1154192037077360e+03
||A*x_1 - b||_2 = 2.4e+02                     1.0e-03[s]
||A*x_2 - b||_2 = 5.9834754277101285e-12                     1.0e-03[s]

Solver: bicgstab with preconditioner: incomplete LU(0) converged in: 28 iterations.
With the stopping criteria "tol * ||Ax_0 - b||_2" is: 1.1e-11

+---------------------------------------------------------+
Total elapsed time:                           1.316e+00[s]
| Preprocessing time:                         3.929e-01[s]
| | Init time:                                1.000e-02[s]
| | Factor time:                              1.208e-01[s]
| Solve time:                                 8.356e-01[s]
| | Iterate time:                             3.463e-03[s]
| | | SpMV time:                              1.190e-02[s]
| | | Precond. time:                          9.077e-01[s]
| | | Dot time:                               2.000e-03[s]
| | | Sum time:                               3.373e-03[s]
| | Sample time:                              8.300e-01[s]
+---------------------------------------------------------+

Device library options in effect: {"trsv_chain": 0, "env": "BIS_TRSV_CHAIN=0"}
"""


def test_cli_output_parser(monkeypatch, tmp_path):
    """_run_cli takes the LAST timer tree of the output, the iteration count, the first / last residual and the options line."""
    fake = tmp_path / "cli.sh"
    fake.write_text("#!/bin/sh\ncat <<'EOT'\n" + CLI_STDOUT + "EOT\n")
    fake.chmod(0o755)
    monkeypatch.setattr(bench, "CLI", str(fake))
    r = bench._run_cli(["unstr:80,80,80", "-bi"], sync_timers=True)
    assert r["iterations"] == 28 and r["converged"] is True
    assert r["solve_s"] == 0.8356 and r["iterate_s"] == 3.463e-03 and r["preprocessing_s"] == 0.3929 and r["factor_s"] == 0.1208
    assert r["spmv_s"] == 1.190e-02 and r["precond_s"] == 0.9077 and r["dot_s"] == 2e-3 and r["sum_s"] == 3.373e-03
    assert r["residual_first"] == 1.1154192037077360e+03 and r["residual_last"] == 5.9834754277101285e-12
    assert json.loads(r["options"]) == {"trsv_chain": 0, "env": "BIS_TRSV_CHAIN=0"}
    # a CLI that fails is an error record, not an exception: the bench line must still be printed
    bad = tmp_path / "bad.sh"
    bad.write_text("#!/bin/sh\necho boom >&2\nexit 3\n")
    bad.chmod(0o755)
    monkeypatch.setattr(bench, "CLI", str(bad))
    assert "boom" in bench._run_cli(["x"], sync_timers=False)["error"]
    monkeypatch.setattr(bench, "CLI", str(tmp_path / "missing"))
    assert "error" in bench._run_cli(["x"], sync_timers=False)


def test_config_legs_use_the_solve_time_of_the_asynchronous_run(monkeypatch, tmp_path):
    fake = tmp_path / "cli.sh"
    fake.write_text("#!/bin/sh\ncat <<'EOT'\n" + CLI_STDOUT + "EOT\n")
    fake.chmod(0o755)
    monkeypatch.setattr(bench, "CLI", str(fake))
    rec = bench.config_legs(only=("config5_unstr_rcm",))["config5_unstr_rcm"]
    assert rec["iterations"] == 28 and rec["solve_s"] == 0.8356 and abs(rec["ms_per_iteration"] - 1e3 * 0.8356 / 28) < 1e-9
    sp = rec["split_with_synchronous_timers"]
    assert abs(sp["blas1_s"] - (2e-3 + 3.373e-3)) < 1e-12 and sp["precond_s"] == 0.9077


def test_sweep_traffic_guard(tmp_path, monkeypatch):
    """PMC bytes per sweep are accepted between 0.9x and 12x the algorithmic bytes, else the record says None."""
    d = tmp_path / "profiles"
    d.mkdir()
    (d / "trsv_traffic.json").write_text(json.dumps({"sweeps": {"k": {"forward": {"hbm_bytes_per_sweep": 2.0e9}, "backward": {"hbm_bytes_per_sweep": 5.0e10}}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.load_sweep_traffic("k", "forward", 1.0e9) == 2.0e9
    assert bench.load_sweep_traffic("k", "forward", 0.25e9) == 2.0e9     # 8x: an ordering without locality (a line per 8-byte operand)
    assert bench.load_sweep_traffic("k", "forward", 0.15e9) is None      # 13x
    assert bench.load_sweep_traffic("k", "backward", 1.0e9) is None     # 50x: not this workload's sweep
    assert bench.load_sweep_traffic("k", "forward", 4.0e9) is None      # fewer bytes than the algorithm needs
    assert bench.load_sweep_traffic("other", "forward", 1.0e9) is None


def test_cpu_topology_and_sptrsv_leg():
    topo = bench.cpu_topology(threads_list=(2,))
    assert {"cgroup_cpu_max", "cgroup_cpu_quota_cores", "numa_nodes", "host_triad", "logical_cpus"} <= set(topo)
    assert topo["host_triad"]["2"]["GBs"] > 0 and 1 <= topo["host_triad"]["2"]["distinct_cpus"] <= 2
    rec = bench.cpu_sptrsv_leg(size=16, seconds=0.2)
    assert rec["kind"] in ("reference", "port") and rec["cores"] == 1
    for direction in ("forward", "backward"):
        assert rec[direction]["ms_per_sweep"] > 0 and rec[direction]["sweeps"] >= 2
