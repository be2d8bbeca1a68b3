"""bench.py's per-rank watchdog (the N > 1 run's guard against a schedule that hangs instead of throwing) and its one-line-on-stdout rule,
without a GPU: the class is plain Python."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(body, **env):
    code = ("import sys, time, json\nsys.path.insert(0, %r)\nimport bench\n" % ROOT) + body
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, **env), cwd=ROOT)


def test_watchdog_prints_the_line_in_hand_and_leaves():
    r = _run("bench.claim_stdout()\nprint('library noise on stdout')\nwd = bench.Watchdog(0)\nwd.mark_measured()\n"
             "wd.set_line({'metric': 'spmm_gflops', 'value': 1.0, 'exchange': 'allgather'})\nwd.arm(\"candidate schedule 'peer2d' (set-up + 2 steps)\", 0.5)\ntime.sleep(60)\nprint('not reached')\n")
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("\n") == 1 and "noise" not in r.stdout            # stdout: the line and nothing else (the noise went to stderr)
    d = json.loads(r.stdout)
    assert d["exchange"] == "allgather" and "peer2d" in d["exchange_watchdog"]["fired_in"]
    assert "watchdog" in r.stderr and "library noise" in r.stderr and "not reached" not in r.stderr


def test_watchdog_exit_codes_and_disarm():
    # nothing measured yet: no line, non-zero
    r = _run("bench.claim_stdout()\nwd = bench.Watchdog(0)\nwd.arm('set-up', 0.3)\ntime.sleep(60)\n")
    assert r.returncode == 3 and r.stdout == ""
    # a rank other than 0 holds no line and still leaves cleanly once the measurement exists
    r = _run("bench.claim_stdout()\nwd = bench.Watchdog(5)\nwd.mark_measured()\nwd.arm('candidate', 0.3)\ntime.sleep(60)\n")
    assert r.returncode == 0 and r.stdout == ""
    # the exit code can be made to read as a failure
    r = _run("bench.claim_stdout()\nwd = bench.Watchdog(0)\nwd.mark_measured()\nwd.set_line({'a': 1})\nwd.arm('candidate', 0.3)\ntime.sleep(60)\n", MI_SPMM_WATCHDOG_EXIT="4")
    assert r.returncode == 4 and json.loads(r.stdout)["a"] == 1
    # disarmed in time: nothing fires; the final line goes out once, by the main thread
    r = _run("bench.claim_stdout()\nwd = bench.Watchdog(0)\nwd.mark_measured()\nwd.set_line({'a': 1})\nwd.arm('candidate', 1.0)\ntime.sleep(0.2)\nwd.disarm()\ntime.sleep(1.5)\n"
             "wd.set_line(None)\nbench.emit_line({'final': True})\nwd.arm('tear-down', 0.3)\ntime.sleep(60)\n")
    assert r.returncode == 0 and r.stdout.count("\n") == 1 and json.loads(r.stdout) == {"final": True}      # a hang in tear-down: the line is out, exit 0
