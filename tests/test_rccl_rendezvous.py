"""The CLI's file rendezvous for the RCCL unique id (host/include/RcclRendezvous.h), on the CPU: the file is unique to the run
(launcher identity in its name), not to a time window — a second run on the same static port cannot read the first run's id,
and a rank that arrives long after rank 0 published still gets it."""
import ctypes as C
import os
import threading
import time

import pytest

from homulator_amd import host


def _lib():
    L = host.load()
    L.hh_rccl_id_path.argtypes = [C.c_char_p, C.c_uint32]
    L.hh_rccl_id_publish.argtypes = [C.c_char_p, C.c_char_p]
    L.hh_rccl_id_fetch.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32]
    L.hh_rccl_id_remove.argtypes = [C.c_char_p]
    return L


def _path(L, env):
    old = {k: os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        buf = C.create_string_buffer(512)
        assert L.hh_rccl_id_path(buf, 512) == 0
        return buf.value.decode()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_name_is_unique_to_the_run():
    L = _lib()
    base = {"HOMULATOR_RCCL_ID_FILE": None, "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29500", "TORCHELASTIC_RUN_ID": None, "TORCHELASTIC_RESTART_COUNT": None}
    a = _path(L, base)
    assert str(os.getppid()) in a and "29500" in a          # the launcher's pid: shared by the ranks of ONE launch
    assert _path(L, {**base, "TORCHELASTIC_RUN_ID": "run-1"}) != _path(L, {**base, "TORCHELASTIC_RUN_ID": "run-2"}) != a
    assert _path(L, {**base, "MASTER_PORT": "29501"}) != a
    # a restarted attempt of the same launcher (same agent pid, same run id) must not find the dead attempt's file
    assert _path(L, {**base, "TORCHELASTIC_RUN_ID": "run-1", "TORCHELASTIC_RESTART_COUNT": "1"}) != _path(L, {**base, "TORCHELASTIC_RUN_ID": "run-1", "TORCHELASTIC_RESTART_COUNT": "0"})
    assert _path(L, {**base, "HOMULATOR_RCCL_ID_FILE": "/tmp/x.id"}) == "/tmp/x.id"


def test_stale_file_and_slow_rank(tmp_path):
    L = _lib()
    p = str(tmp_path / "id").encode()
    stale, fresh = bytes([7]) * 128, bytes(range(128))
    with open(p, "wb") as f:                                 # leftover of an aborted run under the same name
        f.write(stale)
    assert L.hh_rccl_id_remove(p) == 0                       # rank 0 does this before it builds its op
    got = C.create_string_buffer(128)
    assert L.hh_rccl_id_fetch(p, got, 200) != 0              # nothing published yet: a waiting rank does not see the leftover
    assert b"no RCCL id" in L.hh_last_error()
    t = threading.Thread(target=lambda: (time.sleep(0.4), L.hh_rccl_id_publish(p, fresh)))
    t.start()
    assert L.hh_rccl_id_fetch(p, got, 5000) == 0 and got.raw == fresh   # a rank that was already waiting
    t.join()
    os.utime(p, (time.time() - 3600, time.time() - 3600))    # a rank that arrives an hour after rank 0 published
    got2 = C.create_string_buffer(128)
    assert L.hh_rccl_id_fetch(p, got2, 200) == 0 and got2.raw == fresh
    assert L.hh_rccl_id_remove(p) == 0 and not os.path.exists(p)


def test_publish_does_not_follow_a_planted_symlink(tmp_path):
    L = _lib()
    p = str(tmp_path / "id")
    victim = tmp_path / "victim"
    victim.write_bytes(b"precious")
    os.symlink(victim, p + ".tmp")                           # someone else pre-created the temporary name as a link
    assert L.hh_rccl_id_publish(p.encode(), bytes(range(128))) == 0
    assert victim.read_bytes() == b"precious" and not os.path.islink(p)
    assert (os.stat(p).st_mode & 0o777) == 0o600
    os.unlink(p)
    os.symlink(victim, p)                                    # ... or the final name: a waiting rank does not read through it
    got = C.create_string_buffer(128)
    assert L.hh_rccl_id_fetch(p.encode(), got, 100) != 0


def test_library_ops_ignore_the_launcher_environment(monkeypatch):
    """an op built through the library inside a torchrun job (WORLD_SIZE / RANK / LOCAL_RANK set) without `world` is a
    plain one-GPU op on the device it was given; only the CLI infers a rank from the environment"""
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "2")
    monkeypatch.setenv("LOCAL_RANK", "2")
    op = host.Op("config_4_N15.cfg", "hadd", 4, 3, 2, backend=host.BACKEND_COUNT)
    try:
        plan = op.plan()
        assert op.launch_count() >= 1 and not any("EXCH" in l or "replicate" in l for l in plan)
    finally:
        op.close()
