"""Child processes of the test-suite: never without a time limit, and a child that overruns is examined before it dies.

A stuck child once cost a whole GPU run (it blocked `pytest -x` until the box's limit): every test that starts a program
goes through run_child().  When the limit expires the helper
  1. reads what the kernel says about every thread of the child (/proc/PID/task/*: name, state, wait channel, system call),
  2. asks a `taxator` child where it is (SIGUSR1 -> "taxator: stage: ...; engine: ..." on its stderr, host/stage.hh),
  3. kills it (SIGTERM, then SIGKILL) -- a fresh child is never re-executed --
and fails the test BY NAME with all of that in the message.
"""
import os
import signal
import subprocess
import time

import pytest


class ChildResult:
    def __init__(self, returncode, stdout, stderr, seconds):
        self.returncode, self.stdout, self.stderr, self.seconds = returncode, stdout, stderr, seconds


def thread_report(pid):
    """One line per thread of `pid` from /proc (readable for our own children without privileges)."""
    lines = []
    base = "/proc/%d/task" % pid
    try:
        tids = sorted(os.listdir(base), key=int)
    except OSError as e:
        return ["(no /proc entry: %s)" % e]
    for tid in tids:
        def rd(name):
            try:
                with open(os.path.join(base, tid, name)) as f:
                    return f.read().strip()
            except OSError as e:
                return "?(%s)" % e.__class__.__name__
        stat = rd("stat")
        state = stat.rsplit(")", 1)[-1].split()[0] if ")" in stat else "?"
        lines.append("tid %s comm=%s state=%s wchan=%s syscall=%s" % (tid, rd("comm"), state, rd("wchan"), rd("syscall")))
    return lines


def run_child(cmd, timeout=120, stdin_path=None, env=None, cwd=None, what=None):
    """subprocess.run(capture_output=True, text=True) with a limit; on expiry: diagnose, kill, pytest.fail."""
    t0 = time.time()
    stdin = open(stdin_path, "rb") if stdin_path else subprocess.DEVNULL
    try:
        p = subprocess.Popen(cmd, stdin=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=cwd)
    finally:
        if stdin_path:
            stdin.close()
    try:
        out, err = p.communicate(timeout=timeout)
        return ChildResult(p.returncode, out, err, time.time() - t0)
    except subprocess.TimeoutExpired:
        report = thread_report(p.pid)
        try:
            p.send_signal(signal.SIGUSR1)  # taxator: prints its stage and carries on; others: default action ends them
        except OSError:
            pass
        time.sleep(0.5)
        p.terminate()
        try:
            out, err = p.communicate(timeout=5)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
        pytest.fail("%s: child still running after %d s -- killed.\ncommand: %s\nthreads:\n  %s\nstderr (tail):\n%s\nstdout (tail):\n%s" % (
            what or os.path.basename(str(cmd[0])), timeout, " ".join(map(str, cmd)), "\n  ".join(report), (err or "")[-3000:], (out or "")[-1000:]),
            pytrace=False)
