"""One process per GPU without a second communication stack: the launcher and the control channel of `bench.py --gpus N`.

The reference has one parallel region (rayon over tiles, integrator.rs:242-304) inside ONE process; across GPUs the same tile ownership
shards the frame over one process per device (include/shimmer_hip.h "multi-GPU"). The library's RCCL communicator needs exactly one thing
from the host: rank 0's 128-byte unique id carried to every rank. Everything after that (barrier, max-over-ranks clock, counters) runs
through the library's own collectives (shm_dist_barrier / shm_dist_allreduce_f64 / shm_dist_allgather_f64), so a rank process maps one HIP
runtime and one RCCL — the ones libshimmer_hip.so was linked against — and never imports torch.

  FileStore      a directory of small files as the control channel of ONE node: set = write + atomic rename, get = poll. Carries the unique
                 id, per-rank status words (so that a rank that fails before the communicator exists is seen by the others instead of
                 leaving them in ncclCommInitRank), and — in the CPU tests, where no communicator can exist — the barrier and reductions too.
  spawn_ranks    the parent of a plain `python bench.py --gpus N`: starts N fresh children (subprocess, never exec; the parent makes no GPU
                 call) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / SHM_STORE_DIR, relays rank 0's stdout, prefixes
                 the other ranks' output, kills the rest when one fails, and returns non-zero if any child did.
Under `python -m torch.distributed.run` the agent has already started the ranks: store_from_env() then derives the store directory from
MASTER_PORT and the agent's pid, which every worker shares.
"""
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import threading
import time
from pathlib import Path


class StoreTimeout(RuntimeError):
    pass


class FileStore:
    def __init__(self, directory, rank, world, timeout_s=600.0):
        self.dir = Path(directory)
        self.rank, self.world, self.timeout_s = int(rank), int(world), float(timeout_s)
        self.dir.mkdir(parents=True, exist_ok=True)
        self._barriers = 0

    def set(self, key, data: bytes):
        tmp = self.dir / f".{key}.{os.getpid()}.tmp"
        tmp.write_bytes(bytes(data))
        os.replace(tmp, self.dir / key)  # atomic on one filesystem: a reader sees nothing or everything

    def get(self, key, timeout_s=None, abort_on=None):
        """Blocks until `key` exists. abort_on: a key whose appearance means a peer failed (raises with its content)."""
        deadline = time.monotonic() + (self.timeout_s if timeout_s is None else timeout_s)
        f = self.dir / key
        delay = 0.0005
        while True:
            try:
                return f.read_bytes()
            except FileNotFoundError:
                pass
            if abort_on:
                for bad in self.dir.glob(abort_on):
                    raise RuntimeError(f"peer failure reported through the store: {bad.name}: {bad.read_text(errors='replace')[:500]}")
            if time.monotonic() > deadline:
                raise StoreTimeout(f"rank {self.rank}: key {key!r} did not appear in {self.dir} (is every rank running?)")
            time.sleep(delay)
            delay = min(delay * 2, 0.02)

    def fail(self, message):
        """Publishes this rank's failure so that peers waiting on the store stop waiting."""
        try:
            self.set(f"failed.{self.rank}", str(message).encode())
        except OSError:
            pass

    def broadcast(self, key, data, root=0):
        if self.rank == root:
            self.set(key, data)
            return bytes(data)
        return self.get(key, abort_on="failed.*")

    def barrier(self, tag=None):
        self._barriers += 1
        tag = tag or f"barrier{self._barriers}"
        self.set(f"{tag}.{self.rank}", b"1")
        for r in range(self.world):
            self.get(f"{tag}.{r}", abort_on="failed.*")

    def allgather(self, tag, value: bytes):
        self.set(f"{tag}.{self.rank}", value)
        return [self.get(f"{tag}.{r}", abort_on="failed.*") for r in range(self.world)]

    def allreduce_max(self, tag, x: float):
        return max(float(v) for v in self.allgather(tag, repr(float(x)).encode()))

    def finish(self):
        """Last use of the store: a barrier, then every other rank says it will not read again and rank 0 removes the directory."""
        self.barrier("done")
        if self.rank != 0:
            self.set(f"bye.{self.rank}", b"1")
            return
        for r in range(1, self.world):
            self.get(f"bye.{r}", abort_on="failed.*")
        shutil.rmtree(self.dir, ignore_errors=True)


def store_from_env(rank, world, timeout_s=600.0):
    """The store of this launch: SHM_STORE_DIR when spawn_ranks started us; under torch.distributed.run (or any launcher that exports
    RANK / WORLD_SIZE) a directory named after MASTER_PORT and the launching process, which all ranks of one node share."""
    d = os.environ.get("SHM_STORE_DIR")
    if not d:
        port = os.environ.get("MASTER_PORT", "0")
        run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
        # (one directory per ATTEMPT: with --max-restarts the agent re-spawns its workers under the same port / pid / run id, and a new attempt
        #  must not read the dead attempt's rccl_unique_id, failed.* notes or barrier keys)
        attempt = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
        d = os.path.join(tempfile.gettempdir(), f"shm_store_{os.getuid()}_{port}_{os.getppid()}_{run}_a{attempt}")
    return FileStore(d, rank, world, timeout_s)


def _relay(stream, sink, prefix):
    for line in iter(stream.readline, b""):
        sink.write(prefix + line if prefix else line)
        sink.flush()
    stream.close()


def spawn_ranks(argv, n, master_port=None, timeout_s=None, extra_env=None):
    """Starts `n` children running `argv` (one per GPU ordinal 0..n-1) and waits. Returns the exit code for the parent: 0 when every child
    exited 0, otherwise the first non-zero child code (124 on timeout). The parent process makes no GPU call and imports no GPU library."""
    store_dir = tempfile.mkdtemp(prefix="shm_store_")

    def on_signal(signum, frame):  # the launcher is being stopped (a caller's timeout): the ranks must not outlive it
        raise KeyboardInterrupt(f"signal {signum}")
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            signal.signal(sig, on_signal)
        except ValueError:  # not the main thread
            pass
    port = str(master_port or os.environ.get("MASTER_PORT") or (29500 + os.getpid() % 2000))
    procs, threads = [], []
    try:
        for r in range(n):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                       SHM_STORE_DIR=store_dir, SHM_LAUNCHED_BY="shimmer_amd.launch")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the pool's driver only supports dmabuf IPC (RCCL P2P across processes)
            if extra_env:
                env.update(extra_env)
            p = subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
            procs.append(p)
            # rank 0's stdout IS the launcher's stdout (the one JSON line); everything else goes to stderr with a rank prefix
            threads.append(threading.Thread(target=_relay, args=(p.stdout, sys.stdout.buffer if r == 0 else sys.stderr.buffer,
                                                                 b"" if r == 0 else f"[rank {r}] ".encode()), daemon=True))
            threads.append(threading.Thread(target=_relay, args=(p.stderr, sys.stderr.buffer, f"[rank {r}] ".encode()), daemon=True))
        for t in threads:
            t.start()
        deadline = time.monotonic() + timeout_s if timeout_s else None
        rc = 0
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    print(f"[launch] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    _stop(procs, live)
            if live and deadline and time.monotonic() > deadline:
                print(f"[launch] timeout after {timeout_s} s; stopping ranks {sorted(live)}", file=sys.stderr, flush=True)
                _stop(procs, live)
                rc = rc or 124
            if live:
                time.sleep(0.05)
        for t in threads:
            t.join(timeout=5.0)
        return rc
    finally:
        for p in procs:
            if p.poll() is None:
                _kill(p, signal.SIGKILL)
        shutil.rmtree(store_dir, ignore_errors=True)


def _kill(p, sig):
    try:
        os.killpg(p.pid, sig)  # exactly the process group this launcher created for that child (start_new_session)
    except (ProcessLookupError, PermissionError):
        pass


def _stop(procs, live, grace_s=10.0):
    """A rank failed: give the others a moment to see it through the store / the aborted communicator and exit by themselves, then
    terminate what is left."""
    t_end = time.monotonic() + grace_s
    while time.monotonic() < t_end and any(procs[r].poll() is None for r in live):
        time.sleep(0.05)
    for r in live:
        if procs[r].poll() is None:
            _kill(procs[r], signal.SIGTERM)
    t_end = time.monotonic() + 5.0
    while time.monotonic() < t_end and any(procs[r].poll() is None for r in live):
        time.sleep(0.05)
    for r in live:
        if procs[r].poll() is None:
            _kill(procs[r], signal.SIGKILL)


class Watchdog:
    """Ends THIS process (os._exit) if a phase that can block inside a native collective — ncclCommInitRank waits for every rank — does not
    finish in time: a launch must end with an exit code, never hang until the caller's own limit."""

    def __init__(self, seconds, what, store=None):
        self.seconds, self.what, self.store = seconds, what, store
        self._done = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        if not self._done.wait(self.seconds):
            msg = f"watchdog: {self.what} did not finish in {self.seconds} s"
            print(f"[bench] {msg}", file=sys.stderr, flush=True)
            if self.store:
                self.store.fail(msg)
            os._exit(3)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._done.set()
        return False
