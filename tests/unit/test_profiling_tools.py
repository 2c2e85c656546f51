"""The profile post-processing helpers under tools/ (pure Python, no GPU)."""

import csv
import importlib.util
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]


def _load(name: str):
    spec = importlib.util.spec_from_file_location(name, ROOT / "tools" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_short_name_handles_mangled_and_demangled_kernels():
    sp = _load("summarize_pmc")
    mangled = "_ZN12_GLOBAL__N_118conv3x3_fwd_kernelILi2ELi2ELi32EEEvPKDF16bS2_PDF16bPfNS_7ConvGeoE"
    assert sp.short_name(mangled) == "conv3x3_fwd_kernel<2,2,32>"
    assert sp.short_name("_ZN12_GLOBAL__N_120conv3x3_wgrad_kernelEPKDF16bS1_PfNS_7ConvGeoEi") == (
        "conv3x3_wgrad_kernel")
    plain = "(anonymous namespace)::wgrad_reduce_kernel(float const*, bool*, long, int)"
    assert sp.short_name(plain) == "wgrad_reduce_kernel"
    templ = "void (anonymous namespace)::bn_apply_kernel<true, false>(float const*)"
    assert sp.short_name(templ) == "void bn_apply_kernel<true, false>"


def test_steady_state_table_uses_only_the_last_steps(tmp_path):
    trace = tmp_path / "k_kernel_trace.csv"
    rows = []
    t = 0
    # warm-up: a slow library search kernel that must not show up in the table
    for _ in range(5):
        rows.append((t, t + 1_000_000, "naive_conv_search"))
        t += 1_000_100
    for step in range(6):
        rows.append((t, t + 10_000, "conv3x3_fwd_kernel"))
        t += 10_500
        rows.append((t, t + 5_000, "bn_apply_kernel"))
        t += 5_500
        rows.append((t, t + 2_000, "sgd_step_kernel"))
        t += 2_500
    with open(trace, "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["Start_Timestamp", "End_Timestamp", "Kernel_Name"])
        wr.writerows(rows)
    out_json = tmp_path / "table.json"
    res = subprocess.run(
        [sys.executable, str(ROOT / "tools" / "steady_state_kernels.py"), str(trace),
         "--steps", "4", "--json", str(out_json)],
        capture_output=True, text=True, check=True,
    )
    assert "naive_conv_search" not in res.stdout
    table = json.loads(out_json.read_text())
    assert table["steps"] == 4
    by_name = {k["kernel"]: k for k in table["kernels"]}
    assert by_name["conv3x3_fwd_kernel"]["calls_per_step"] == 1.0
    assert by_name["conv3x3_fwd_kernel"]["us_per_call"] == 10.0
    assert abs(table["busy_us_per_step"] - 17.0) < 1e-6


def _trace_db(path, dispatches):
    """A result database with the two tables the trace tools read."""
    import sqlite3

    con = sqlite3.connect(path)
    con.execute("create table rocpd_info_kernel_symbol (id integer, kernel_name text)")
    con.execute("create table rocpd_kernel_dispatch (queue_id integer, start integer, end integer, "
                "kernel_id integer)")
    names = sorted({d[3] for d in dispatches})
    con.executemany("insert into rocpd_info_kernel_symbol values (?, ?)", list(enumerate(names)))
    con.executemany("insert into rocpd_kernel_dispatch values (?, ?, ?, ?)",
                    [(q, s, e, names.index(n)) for q, s, e, n in dispatches])
    con.commit()
    con.close()


def test_idle_intervals_count_time_no_queue_is_busy(tmp_path):
    # two queues: queue 1 covers the first gap of queue 0 but not the second
    ns = 1000
    dispatches = [
        (0, 0, 100 * ns, "a_kernel"), (0, 150 * ns, 250 * ns, "b_kernel"),
        (1, 90 * ns, 160 * ns, "c_kernel"),
        (0, 300 * ns, 400 * ns, "a_kernel"),
    ]
    db = tmp_path / "t_results.db"
    _trace_db(db, dispatches)
    res = subprocess.run(
        [sys.executable, "idle_intervals.py", str(db), "--min-us", "10"],
        capture_output=True, text=True, check=True, cwd=ROOT / "tools",
    )
    lines = res.stdout.splitlines()
    assert lines[0] == "span_us,400.0" and lines[1] == "idle_us,50.0"
    assert lines[2] == "busy_share,0.8750"
    assert "b_kernel,a_kernel,1,50.0,50.0,50.0" in lines
    assert not any(line.startswith("a_kernel,b_kernel") for line in lines)
