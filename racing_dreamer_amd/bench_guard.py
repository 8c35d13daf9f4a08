"""The parts of bench.py that must work when something else does not: the launcher of an N > 1 run and the guard that
prints rank 0's ONE JSON line whatever happens.  No torch, no GPU: importable (and tested) on a CPU-only box.

Exit codes of `python bench.py` (the line on stdout says the same in words):

    0    the line is complete: the headline and every leg that was started
    3    EXIT_HEADLINE_MISSING   the run ended before the headline had been measured (a hang or an error in the rendezvous, the
                                 ring prefill or the timed window): the line printed is PROVISIONAL (`headline_pending: true`,
                                 rank 0's own simulation-only window, no collective had completed) - not a measurement of N GPUs
    4    EXIT_CHECK_MISMATCH     a gathered record of the HEADLINE payload differs from what its sender sent
    5    EXIT_LEG_LOST           the headline is measured and in the line, but a leg after it raised or hung (`aborted` /
                                 `leg_errors` name it); legs skipped because the time budget ran out do NOT count (`legs_skipped`)
    124  EXIT_LAUNCH_TIMEOUT     the ranks the self-launcher started did not finish within --launch-timeout; all of them killed
    128 + s                      the launcher (or a rank) was ended by signal s; every rank it had started was ended first
"""
from __future__ import annotations

import contextlib
import ctypes
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

EXIT_OK, EXIT_HEADLINE_MISSING, EXIT_CHECK_MISMATCH, EXIT_LEG_LOST, EXIT_LAUNCH_TIMEOUT = 0, 3, 4, 5, 124
T0_ENV = "RC_BENCH_T0"               # epoch seconds at which the outermost bench.py process started: every rank budgets from it
_PR_SET_PDEATHSIG = 1


def die_with_parent(sig=signal.SIGTERM):
    """Ask the kernel to send `sig` to THIS process when the thread that started it dies (prctl PR_SET_PDEATHSIG): a
    launcher that was killed outright (SIGKILL: no handler runs) still takes its ranks with it."""
    try:
        ctypes.CDLL(None, use_errno=True).prctl(_PR_SET_PDEATHSIG, int(sig), 0, 0, 0)
    except Exception:                                   # noqa: BLE001 - not Linux / no libc: the signal handlers remain
        pass


def _descendants(pid):
    """PIDs of every live descendant of `pid` (by /proc; psutil if present)."""
    try:
        import psutil
        return [p.pid for p in psutil.Process(pid).children(recursive=True)]
    except Exception:                                   # noqa: BLE001
        pass
    kids, todo = [], [pid]
    while todo:
        p = todo.pop()
        try:
            for t in os.listdir(f"/proc/{p}/task"):
                with open(f"/proc/{p}/task/{t}/children") as f:
                    for c in f.read().split():
                        kids.append(int(c))
                        todo.append(int(c))
        except OSError:
            pass
    return kids


def _alive(pid):
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"
    except OSError:
        return False


class _Ended(Exception):
    def __init__(self, signum):
        self.signum = signum


def end_process_trees(children, grace_s=20.0, log=None):
    """End every process in `children` (Popen objects, each started as the leader of a session of its own) and everything
    below them: SIGTERM to each one's process group - the ranks print their line and leave - then, after `grace_s`, SIGKILL
    to the groups and to every descendant seen before the first signal, by exact PID (a process that moved itself into
    another group is still found).  Returns the PIDs that had to be killed the hard way."""
    pids = []
    for c in children:
        pids += [c.pid] + _descendants(c.pid)
    for c in children:
        try:
            os.killpg(c.pid, signal.SIGTERM)
        except (ProcessLookupError, PermissionError):
            pass
    t_end = time.monotonic() + grace_s
    while time.monotonic() < t_end:
        for c in children:
            c.poll()
        if not any(_alive(p) for p in pids):
            break
        time.sleep(0.1)
    hard = [p for p in pids if _alive(p)]
    if hard:
        if log:
            log(f"{len(hard)} process(es) still alive {grace_s:.0f} s after SIGTERM - SIGKILL: {hard}")
        for c in children:
            try:
                os.killpg(c.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
        for p in hard:
            try:
                os.kill(p, signal.SIGKILL)
            except ProcessLookupError:
                pass
    for c in children:
        try:
            c.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pass
    return hard


def self_launch(n_ranks, timeout_s, script, argv, t0=None, straggler_s=30.0):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves - N children running
    `python bench.py <same arguments>` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set (the
    env:// rendezvous of torch.distributed; what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would set),
    each in a session of its own - BEFORE this process has touched the GPU (nothing here imports torch), pass their output
    through (rank 0 prints the JSON line), and exit with the ranks' own exit code (rank 0's if it failed, else the first
    non-zero one: the codes at the top of this file reach the caller as they are).  What can go wrong outside the ranks is
    handled HERE:

    * SIGTERM / SIGINT / SIGHUP to this process (a driver's time-out ends the process it started, not the sessions the ranks
      live in): the handler forwards SIGTERM to every rank's process group, so rank 0's guard prints the line it has, waits
      for every descendant to be gone, kills what is left, and exits 128 + signal.  No rank survives its launcher;
    * this process killed outright: every rank was started with PR_SET_PDEATHSIG (and sets it again for itself), so the kernel
      delivers the SIGTERM no handler could;
    * a rank that has left while others go on (it failed; the others may sit in a collective it will never join): the
      others get `straggler_s` to leave by themselves - the guard's key-value store tells them why - then the same teardown;
    * the ranks not finishing within `timeout_s` of the start (default 540 s: inside the driver's 600 s): teardown, exit 124.
    Never re-executes a process that has initialised the GPU."""
    t0 = time.time() if t0 is None else t0
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    base = dict(os.environ)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL and hipIpc* need on this driver
    base.setdefault("OMP_NUM_THREADS", "1")
    base.update({T0_ENV: repr(t0), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks), "MASTER_ADDR": "127.0.0.1",
                 "MASTER_PORT": str(port), "GROUP_RANK": "0", "ROLE_RANK": "0"})

    def log(msg):
        print(f"bench.py launcher: {msg}", file=sys.stderr, flush=True)

    def on_signal(signum, _frame):
        raise _Ended(signum)

    handled = []
    for s in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            signal.signal(s, on_signal)
            handled.append(s)
        except (ValueError, OSError):
            pass

    def quiet():
        for s in handled:
            signal.signal(s, signal.SIG_IGN)               # (a second signal must not interrupt the teardown)

    ranks = []
    try:
        for r in range(n_ranks):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            ranks.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                          start_new_session=True, preexec_fn=die_with_parent))
        first_exit = None
        while True:
            codes = [p.poll() for p in ranks]
            if all(c is not None for c in codes):
                break
            now = time.time()
            if now - t0 > timeout_s:
                raise subprocess.TimeoutExpired("bench.py ranks", timeout_s)
            if any(c not in (None, 0) for c in codes):
                first_exit = first_exit or now
                if now - first_exit > straggler_s:
                    gone = [r for r, c in enumerate(codes) if c is not None]
                    log(f"rank(s) {gone} left {straggler_s:.0f} s ago, the others are still running - ending them")
                    quiet()
                    end_process_trees([p for p in ranks if p.poll() is None], log=log)
                    break
            time.sleep(0.2)
        codes = [p.poll() for p in ranks]
        rc = codes[0] if codes[0] else next((c for c in codes if c), 0)
        if rc is None or rc < 0:
            rc = 128 - rc if rc else 1                      # (a rank ended by a signal)
    except subprocess.TimeoutExpired:
        log(f"the {n_ranks} ranks did not finish within {timeout_s:.0f} s of the start - ending them")
        quiet()
        end_process_trees(ranks, log=log)
        rc = EXIT_LAUNCH_TIMEOUT
    except _Ended as e:
        quiet()
        log(f"signal {e.signum} - forwarding SIGTERM to the {len(ranks)} ranks and waiting for them")
        end_process_trees(ranks, log=log)
        rc = 128 + e.signum
    sys.exit(rc)


class LineGuard:
    """Rank 0 prints exactly ONE JSON line, whatever happens - from the moment there is anything to print.

    N > 1 runs in three stages.  (1) Every rank times a simulation-only window of its own - no collective, nothing a peer can
    hold up - and rank 0 arms the guard with a PROVISIONAL line built from it (`arm(line, pending=True)`: `headline_pending`).
    (2) The first collectives this code ever issues across devices - the rendezvous, the ring prefill, the timed headline
    window - run as legs with deadlines of their own; if one of them hangs or raises, rank 0 prints the provisional line with
    `aborted` and every rank exits EXIT_HEADLINE_MISSING (3).  (3) `promote(line)` replaces the provisional line by the
    measured headline; from then on a leg that raises or overruns ends the run with that line + `aborted`, exit EXIT_LEG_LOST
    (5) - a hung collective cannot be recovered in-process, and a rank that raised has left the others' collective sequence
    (it says so through the job's key-value store, so the others do not wait for their own deadline).  At N = 1 a failed leg
    is recorded (`leg_errors`), the run goes on - there is nobody to fall out of step with - and the exit code at the end is 5.

    Time: `deadline` (epoch seconds) is when the whole run must be over - `--time-budget` after the OUTERMOST process
    started (the launcher hands its start time down in RC_BENCH_T0).  A leg's own deadline is the smaller of `--leg-timeout`
    and what is left of the budget minus a reserve for printing and leaving; `go(name, need_s)` says whether a leg is worth
    starting at all (rank 0 decides by its clock and the others read the decision from the store: every rank takes the same
    branch), and a leg that is skipped is listed in `legs_skipped` - not an error.

    SIGTERM / SIGINT (installed by `install()`, before anything is armed): the handler only raises a flag; the watchdog
    thread prints the line - if there is one - and ends the process with 128 + signal."""

    KEY = "rc_bench_abort"
    RESERVE_S = 12.0

    def __init__(self, rank, world, timeout_s, deadline=None):
        self.rank, self.world, self.timeout = rank, world, float(timeout_s)
        self.run_deadline = deadline            # epoch seconds or None
        self.line, self.store = None, None
        self.lock = threading.RLock()
        self.printed = False
        self.leg_name, self.deadline = None, None
        self.errors, self.skipped, self.leg_seconds = {}, {}, {}
        self._stop = False
        self._signal = None
        self._watching = False
        self._go_no = 0
        self._leg_budget = self.timeout
        self.armed = False
        self.pending = False

    # ------------------------------------------------------------------ set-up
    def install(self):
        """Signal handlers and the watchdog thread - before anything is armed, so that a run ended from outside during the
        set-up leaves through the same door (there is no line to print yet: it says so on stderr)."""
        for s in (signal.SIGTERM, signal.SIGINT):
            try:
                # the handler runs in the main thread, possibly while that thread is inside emit() / finalise() holding the
                # lock: it only raises a flag, the watchdog thread prints the line
                signal.signal(s, lambda signum, _f: setattr(self, "_signal", signum))
            except (ValueError, OSError):
                pass
        self._watch_start()

    def _watch_start(self):
        if not self._watching:
            self._watching = True
            threading.Thread(target=self._watch, daemon=True).start()

    def arm(self, line, store=None, pending=False):
        self.line, self.armed, self.pending = line, True, bool(pending)
        if store is not None:
            self.store = store
        if pending and line is not None:
            line["headline_pending"] = True
        if not self._watching:
            self.install()

    def set_store(self, store):
        self.store = store

    def promote(self, line):
        """The measured headline replaces the provisional line."""
        with self.lock:
            self.line, self.pending, self.armed = line, False, True

    # ------------------------------------------------------------------ time
    def time_left(self):
        return float("inf") if self.run_deadline is None else self.run_deadline - time.time()

    def budget(self, want_s=None):
        """Seconds a leg starting now may take: `want_s` (default --leg-timeout), capped by what is left of the run."""
        want = self.timeout if want_s is None else float(want_s)
        return max(min(want, self.time_left() - self.RESERVE_S), 1.0)

    def go(self, name, need_s, kind=None, weight=1.0):
        """Is there time for leg `name`, which needs about `need_s` seconds - or, if that is more, 1.3 x what the legs of the same
        `kind` so far took per unit of `weight` (for the payload legs: bytes per step x steps, so a record twice as large is
        expected to take twice as long), times this leg's weight?  The same answer on every rank."""
        i, self._go_no = self._go_no, self._go_no + 1
        if kind is not None:
            seen = [sec / w for k, (sec, w) in self.leg_seconds.items() if k.startswith(kind + ":")]
            if seen:
                need_s = max(need_s, 1.3 * max(seen) * weight)
        key = f"rc_bench_go_{i}"
        if self.world > 1 and self.store is not None and self.rank != 0:
            try:
                import datetime
                self.store.wait([key], datetime.timedelta(seconds=90))
                ok = self.store.get(key) == b"1"
            except Exception as exc:                       # noqa: BLE001 - rank 0 is gone or silent: leave through the guard
                self.finalise(name, f"rank {self.rank}: no decision from rank 0 on leg {name!r} ({type(exc).__name__})")
                ok = False
        else:
            ok = self.time_left() - self.RESERVE_S >= need_s
            if self.world > 1 and self.store is not None:
                self.store.set(key, "1" if ok else "0")
        if not ok:
            self.skipped[name] = f"needs about {need_s:.0f} s, {max(self.time_left(), 0.0):.0f} s of the time budget were left"
            if self.rank == 0:
                print(f"bench.py: leg {name!r} skipped: {self.skipped[name]}", file=sys.stderr, flush=True)
        return ok

    # ------------------------------------------------------------------ the watchdog
    def _store_reason(self):
        if self.store is None:
            return None
        try:
            if self.store.check([self.KEY]):
                return self.store.get(self.KEY).decode(errors="replace")
        except Exception as exc:                       # the store lives in another process: gone = the job is ending
            return f"key-value store unreachable ({type(exc).__name__})"
        return None

    def _watch(self):
        while not self._stop:
            time.sleep(0.25)
            if self._signal is not None:
                self.finalise("signal", f"signal {self._signal}", code=128 + int(self._signal))
            d, name = self.deadline, self.leg_name
            if d is not None and time.monotonic() > d and name == "shutdown":
                # the line is out and complete; a rank that never reaches the closing barrier must not turn that into a time-out
                self.finalise(name, f"the shutdown took longer than {self._leg_budget:.0f} s", code=self.exit_code())
            if d is not None and time.monotonic() > d:
                self.finalise(name, f"exceeded its deadline of {self._leg_budget:.0f} s"
                                    + (f" ({max(self.time_left(), 0.0):.0f} s of the run's time budget left)" if self.run_deadline else ""))
            why = self._store_reason() if (self.world > 1 and self.armed) else None
            if why:
                time.sleep(0.3)
                self.finalise(*(why.split("|", 1) if "|" in why else (self.leg_name, why)))

    def finalise(self, leg, reason, code=None):
        with self.lock:
            if code is None:
                code = EXIT_HEADLINE_MISSING if (self.pending or not self.armed) else EXIT_LEG_LOST
            if self.rank == 0 and self.line is not None and not self.printed:
                note = ("the run ended BEFORE the headline was measured: `value` is rank 0's own simulation-only window times the number "
                        "of ranks, no collective across the ranks had completed - not a measurement of this many GPUs"
                        if self.pending else
                        "the run was ended after the headline leg: everything above was measured; legs that had not run are absent")
                self.line["aborted"] = {"leg": leg, "reason": reason, "exit_code": code, "note": note}
                self._decorate()
                print(json.dumps(self.line), flush=True)
                self.printed = True
            print(f"bench.py: rank {self.rank}: run ended in leg {leg!r}: {reason} (exit {code})", file=sys.stderr, flush=True)
            sys.stdout.flush()
            os._exit(code)

    def _decorate(self):
        if self.errors:
            self.line["leg_errors"] = self.errors
        if self.skipped:
            self.line["legs_skipped"] = self.skipped

    # ------------------------------------------------------------------ legs
    @contextlib.contextmanager
    def leg(self, name, budget_s=None, kind=None, weight=1.0):
        self._leg_budget = self.budget(budget_s)
        self.leg_name, self.deadline = name, (time.monotonic() + self._leg_budget if self.armed else None)
        t_in = time.monotonic()
        hook = os.environ.get("RC_BENCH_FAIL_LEG", "").split(":")          # tests: "<leg>[:<rank>]" raises, "RC_BENCH_HANG_LEG" sleeps
        hang = os.environ.get("RC_BENCH_HANG_LEG", "").split(":")
        try:
            if hook[0] == name and (len(hook) < 2 or int(hook[1]) == self.rank):
                raise RuntimeError("RC_BENCH_FAIL_LEG")
            if hang[0] == name and (len(hang) < 2 or int(hang[1]) == self.rank):
                time.sleep(1e6)
            yield
            self.leg_seconds[(kind + ":" if kind else "") + name] = (time.monotonic() - t_in, float(weight))
        except Exception as exc:                           # noqa: BLE001 - every failure of a guarded leg is data
            msg = f"{type(exc).__name__}: {exc}"
            self.errors[name] = msg
            print(f"bench.py: rank {self.rank}: leg {name!r} failed: {msg}", file=sys.stderr, flush=True)
            if not self.armed:
                raise                                      # (not armed: N = 1's headline itself - an exception is an exception)
            if self.world > 1 or self.pending:
                first = self._store_reason() if self.world > 1 else None      # another rank failed first: this exception is its echo (a peer that left)
                if first and "|" in first:
                    self.finalise(*first.split("|", 1))
                try:
                    if self.store is not None:
                        self.store.set(self.KEY, f"{name}|rank {self.rank}: {msg}")
                except Exception:                          # noqa: BLE001
                    pass
                time.sleep(2.0 if self.store is not None else 0.0)       # let the others read the reason before this rank's exit breaks their collective
                self.finalise(name, f"rank {self.rank}: {msg}")
        finally:
            self.leg_name, self.deadline = None, None

    def emit(self, shutdown_s=30.0):
        """Print the line (rank 0, once).  The watchdog stays on for the shutdown that follows - the closing barrier, the
        process group's destruction: a rank that never arrives there must not turn a finished run into a time-out - and ends
        the process (with the code the line deserves) if that takes longer than `shutdown_s`; `done()` switches it off."""
        with self.lock:
            if self.rank == 0 and not self.printed and self.line is not None:
                self._decorate()
                print(json.dumps(self.line), flush=True)
                self.printed = True
            self._leg_budget = shutdown_s
            self.leg_name, self.deadline = "shutdown", time.monotonic() + shutdown_s

    def exit_code(self):
        return EXIT_LEG_LOST if self.errors else EXIT_OK

    def done(self):
        self._stop = True
        self.deadline = None
