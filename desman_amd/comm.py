"""RCCL communicator of the library (include/desman_hip.h: dsm_comm_*), no torch: one per process / GPU.

What it carries: the chain scheduler's single exchange -- one fixed-size fit record per chain, all-gathered at the end (the
reference "gathers" with `cat */fit.txt`, complete_example/README.md:626-627, after the N background jobs of
scripts/runDesman.sh:15-21) -- and the per-iteration all-reduce of a chain sharded by positions (desman_amd/vshard.py).

Bootstrap: RCCL needs one 128-byte id, made by rank 0, in every rank's hands.  Rank 0 serves it on a TCP socket at
(MASTER_ADDR, DESMAN_COMM_PORT or MASTER_PORT + 17); the other ranks fetch it.  The launcher (desman_amd/launch.py or
torch.distributed.run) provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.
"""
import ctypes as C
import os
import socket
import time

import numpy as np

from . import _lib

ID_BYTES = 128


def _port():
    if os.environ.get("DESMAN_COMM_PORT"):
        return int(os.environ["DESMAN_COMM_PORT"])
    return int(os.environ.get("MASTER_PORT", "29500")) + 17


def _serve_id(uid, world, addr, port, timeout):
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind((addr, port))
    srv.listen(world)
    srv.settimeout(timeout)
    try:
        for _ in range(world - 1):
            conn, _ = srv.accept()
            with conn:
                conn.sendall(uid)
    finally:
        srv.close()


def _fetch_id(addr, port, timeout):
    t_end = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as s:
                buf = b""
                while len(buf) < ID_BYTES:
                    chunk = s.recv(ID_BYTES - len(buf))
                    if not chunk:
                        break
                    buf += chunk
                if len(buf) == ID_BYTES:
                    return buf
        except OSError:
            pass
        if time.time() > t_end:
            raise _lib.DesmanHipError("comm bootstrap: no unique id from rank 0 at %s:%d within %.0f s" % (addr, port, timeout))
        time.sleep(0.05)


class Comm:
    """`Comm.from_env()` in every rank of a launch; then allgather / allreduce / barrier on numpy float64 arrays"""

    def __init__(self, rank, world, device, uid):
        self.lib = _lib.load()
        self._h = C.c_void_p()
        buf = C.create_string_buffer(uid, ID_BYTES)
        _lib.check(self.lib.dsm_comm_create(C.byref(self._h), C.cast(buf, C.c_void_p), int(rank), int(world), int(device)))
        self.rank, self.world, self.device = int(rank), int(world), int(device)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(ID_BYTES)
        _lib.check(_lib.load().dsm_comm_unique_id(C.cast(buf, C.c_void_p)))
        return buf.raw

    @classmethod
    def from_env(cls, device=None, timeout=300.0):
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        if rank == 0:
            uid = cls.unique_id()
            if world > 1:
                _serve_id(uid, world, addr, _port(), timeout)
        else:
            uid = _fetch_id(addr, _port(), timeout)
        return cls(rank, world, local if device is None else device, uid)

    def allgather(self, a):
        """[world, n] <- every rank's vector a[n]"""
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        out = np.empty((self.world, a.size))
        if a.size:
            _lib.check(self.lib.dsm_comm_allgather_f64(self._h, a, out.reshape(-1), a.size))
        return out

    def allreduce(self, a, op="sum"):
        a = np.ascontiguousarray(a, dtype=np.float64).copy()
        flat = a.reshape(-1)
        _lib.check(self.lib.dsm_comm_allreduce_f64(self._h, flat.ctypes.data, flat.size, {"sum": 0, "max": 1}[op]))
        return a

    def barrier(self):
        _lib.check(self.lib.dsm_comm_barrier(self._h))

    def close(self):
        if self._h:
            self.lib.dsm_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:                                  # noqa: BLE001
            pass
