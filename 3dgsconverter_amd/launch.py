"""One process per GPU without torch: rank discovery, spawning, and the hand-over of the communicator's unique id.

The reference is a single process; this is launcher plumbing of its MI355X scale-out.  Two ways in:

  * a launcher already started the ranks (``python -m torch.distributed.run --nproc-per-node N ...`` or any other that
    sets RANK / LOCAL_RANK / WORLD_SIZE): ``rank_env()`` reads them; the 128-byte id of ``gsx_comm_unique_id`` travels
    through a small file that rank 0 writes atomically and the others poll (one node: /tmp is shared) -- torch's TCP store
    is not needed and torch is not imported;
  * nobody did: ``spawn_ranks()`` starts N copies of the calling script with those variables set (``bench.py --gpus N``
    run plainly).

Device per rank: ``LOCAL_RANK`` when the node shows at least WORLD_SIZE GPUs (every rank sees every GPU, as under torchrun:
RCCL needs the peers visible), otherwise ``LOCAL_RANK % device_count`` -- ranks then SHARE GPUs, which RCCL refuses, and the
communicator uses the shared-memory "hostwire" transport (csrc/comm.hip): a functional run of the same code, not a scaling
measurement.  The ranks then settle the transport among themselves from the GPUs they actually opened
(``agree_transport``: PCI bus ids, all distinct -> RCCL), which also covers launchers that show each rank one GPU.
``GSX_COMM_TRANSPORT=hostwire|rccl`` overrides the choice.
"""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile
import time


def rank_env():
    """(rank, local_rank, world) from the launcher's environment; (0, 0, 1) when there is none"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, local, world


def _parent_tag() -> str:
    """names the job for ranks started by one launcher process: its pid and start time (a recycled pid differs in the latter)"""
    ppid = os.getppid()
    start = "0"
    try:
        with open("/proc/%d/stat" % ppid) as f:
            start = f.read().rsplit(")", 1)[1].split()[19]   # field 22: starttime
    except Exception:   # noqa: BLE001 -- no /proc: the pid and port alone
        pass
    return "%d_%s" % (ppid, start)


def _rendezvous_dir() -> str:
    """a directory only this user can enter (0700, owned by us, not a symlink): the id and device files of a job are not
    plantable or readable by another local user"""
    d = os.path.join(tempfile.gettempdir(), "gsx_rdzv_u%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError("rendezvous directory %s is not a private directory of uid %d" % (d, os.getuid()))
    return d


def rendezvous_path() -> str:
    """the file the unique id travels through: GSX_RDZV_FILE (spawn_ranks sets it) or a name every rank of one launcher
    derives identically (parent process identity + MASTER_PORT + torchrun's run id + its restart count: workers restarted
    by the same launcher must not meet the id file of the attempt that died)"""
    p = os.environ.get("GSX_RDZV_FILE")
    if p:
        return p
    tag = "%s_%s_%s_r%s" % (_parent_tag(), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                            os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    return os.path.join(_rendezvous_dir(), "".join(ch if ch.isalnum() or ch in "_-" else "_" for ch in tag))


def _write_private(path: str, data: bytes):
    """atomic publish: an exclusive, no-follow temporary (a pre-placed symlink or file makes it fail, not follow), renamed"""
    tmp = "%s.tmp%d" % (path, os.getpid())
    try:
        os.unlink(tmp)
    except OSError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    try:
        os.write(fd, data)
        os.fsync(fd)
    finally:
        os.close(fd)
    os.replace(tmp, path)   # atomic: a reader sees nothing or everything


def _read_private(path: str):
    """bytes of a file of OURS, or None while it does not exist (never follows a symlink)"""
    try:
        fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    except FileNotFoundError:
        return None
    try:
        if os.fstat(fd).st_uid != os.getuid():
            raise PermissionError("%s is not owned by uid %d" % (path, os.getuid()))
        return os.read(fd, 4096)
    finally:
        os.close(fd)


def exchange_unique_id(rank: int, make_id, path: str | None = None, timeout_s: float = 300.0) -> bytes:
    """rank 0 calls make_id() -> 128 bytes and publishes them; every rank returns the same bytes"""
    path = path or rendezvous_path()
    if rank == 0:
        uid = make_id()
        _write_private(path, uid)
        return uid
    t0 = time.time()
    while True:
        uid = _read_private(path)
        if uid is not None and len(uid) == 128:
            return uid
        if time.time() - t0 > timeout_s:
            raise TimeoutError("rank %d: no unique id at %s after %.0f s (did rank 0 start?)" % (rank, path, timeout_s))
        time.sleep(0.01)


def retire_unique_id(rank: int, path: str | None = None):
    """after every rank has initialised its communicator (a barrier later): the file has done its job"""
    if rank == 0:
        path = path or rendezvous_path()
        for name in [path] + ["%s.dev%d" % (path, r) for r in range(int(os.environ.get("WORLD_SIZE", "1")))]:
            try:
                os.unlink(name)
            except OSError:
                pass


def pick_device_and_transport(local_rank: int, world: int, device_count: int):
    """-> (device index, transport name).  The transport is a first guess from the device count alone; agree_transport()
    settles it from the devices the ranks actually opened."""
    forced = os.environ.get("GSX_COMM_TRANSPORT")
    if device_count >= world:
        return local_rank, forced or "rccl"
    return local_rank % max(device_count, 1), forced or "hostwire"


def agree_transport(rank: int, world: int, device_uid: str, path: str | None = None, timeout_s: float = 300.0) -> str:
    """every rank publishes the identity of the GPU it opened (``_lib.device_uid``: the PCI bus id) next to the rendezvous
    file and reads the others': all different -> "rccl", any two equal -> "hostwire".  This also covers launchers that
    give each rank ONE visible GPU (HIP_VISIBLE_DEVICES per rank: device_count() is 1 on eight different GPUs).
    ``GSX_COMM_TRANSPORT`` still overrides."""
    forced = os.environ.get("GSX_COMM_TRANSPORT")
    if forced:
        return forced
    path = path or rendezvous_path()
    _write_private("%s.dev%d" % (path, rank), (device_uid + "\n").encode())
    seen, t0 = {}, time.time()
    while len(seen) < world:
        for r in range(world):
            if r in seen:
                continue
            raw = _read_private("%s.dev%d" % (path, r))
            txt = raw.decode() if raw is not None else ""
            if txt.endswith("\n"):
                seen[r] = txt.strip()
        if len(seen) < world:
            if time.time() - t0 > timeout_s:
                raise TimeoutError("rank %d: %d of %d ranks announced their device at %s.dev*" % (rank, len(seen), world, path))
            time.sleep(0.01)
    return "rccl" if len(set(seen.values())) == world else "hostwire"


class comm_watchdog:
    """A multi-process job must end with a line its driver can parse, not with the driver's own timeout: if a STAGE of the
    run (communicator set-up, a timed region, ...) does not finish within `timeout_s`, rank 0 writes ONE JSON line with an
    "error" field to `json_fd` and every rank leaves with exit code 124.  The blocking calls are ctypes calls into librccl /
    libgsx_hip, which release the GIL, so this thread runs while the main one is stuck.  stage(name) restarts the clock."""

    def __init__(self, rank: int, world: int, json_fd: int, timeout_s: float, fields: dict | None = None):
        import threading
        self.rank, self.world, self.json_fd, self.timeout_s = rank, world, json_fd, float(timeout_s)
        self.fields = dict(fields or {})
        self._stage, self._t0, self._done = "communicator set-up", time.time(), threading.Event()
        self._lock = threading.Lock()
        if self.timeout_s > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def stage(self, name: str):
        with self._lock:
            self._stage, self._t0 = name, time.time()

    def done(self):
        self._done.set()

    def _run(self):
        while not self._done.wait(0.25):
            with self._lock:
                late, stage = time.time() - self._t0 > self.timeout_s, self._stage
            if late:
                msg = "rank %d of %d: no progress for %.0f s after stage '%s' (GSX_COMM_TIMEOUT)" % (self.rank, self.world, self.timeout_s, stage)
                sys.stderr.write("[gsx launch] " + msg + "\n")
                if self.rank == 0:
                    import json
                    line = dict(self.fields, value=None, ms_per_step=None, vs_baseline=None, error=msg)
                    try:
                        os.write(self.json_fd, (json.dumps(line) + "\n").encode())
                    except OSError:
                        pass
                os._exit(124)


def spawn_ranks(world: int, argv=None, env_extra=None, timeout_s: float | None = None) -> int:
    """start `world` copies of the calling script (same arguments), one rank each, and wait for them.  stdout / stderr are
    inherited (only rank 0 prints the result line).  -> the first non-zero exit code, else 0"""
    argv = list(sys.argv if argv is None else argv)
    rdzv_dir = tempfile.mkdtemp(prefix="gsx_job_")
    procs = []
    try:
        for r in range(world):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                        "GSX_RDZV_FILE": os.path.join(rdzv_dir, "unique_id"), "GSX_SPAWNED": "1"})
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)
            if env_extra:
                env.update(env_extra)
            procs.append(subprocess.Popen([sys.executable] + argv, env=env))
        rc, t0 = 0, time.time()
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:      # a rank died: the others would wait for it in a collective
                        q.terminate()
            if timeout_s is not None and time.time() - t0 > timeout_s:
                for q in pending:
                    q.kill()
                return 124
            time.sleep(0.02)
        return rc
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            for f in os.listdir(rdzv_dir):
                os.unlink(os.path.join(rdzv_dir, f))
            os.rmdir(rdzv_dir)
        except OSError:
            pass
