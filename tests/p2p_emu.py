"""Host mirror of csrc/p2p.hip's exchange protocol (test infrastructure): the same buffer layout
([2 parities][world slots][slot] doubles + [2 parities][world] flags), the same steps in the same order -- advance the
sequence number, write my slot of EVERY rank's buffer, publish the number, wait for all ranks' numbers in my own buffer,
sum the slots in rank order -- over POSIX shared memory instead of peer-mapped device memory.  What it pins on the CPU:
that two parities suffice (a rank may run ahead by at most one collective), that results are identical on every rank,
and the host-side bookkeeping of semseg_amd/p2p.py's contract (message <= slot)."""
import time
from multiprocessing import shared_memory

import numpy as np


class HostP2P:
    def __init__(self, rank, world, slot, names, create=False):
        self.rank, self.world, self.slot = rank, world, slot
        self.nbytes = 2 * world * slot * 8 + 2 * world * 64
        self.shms = []
        for r, name in enumerate(names):
            if r == rank and create:
                shm = shared_memory.SharedMemory(name=name, create=True, size=self.nbytes)
                shm.buf[:self.nbytes] = bytes(self.nbytes)
            else:
                shm = None
            self.shms.append(shm)
        self.names = names
        self.seq = 0

    def attach(self):
        for r, name in enumerate(self.names):
            if self.shms[r] is None:
                self.shms[r] = shared_memory.SharedMemory(name=name)
        self.data = [np.ndarray((2, self.world, self.slot), dtype=np.float64, buffer=s.buf) for s in self.shms]
        self.flags = [np.ndarray((2, self.world, 8), dtype=np.uint64, buffer=s.buf, offset=2 * self.world * self.slot * 8)
                      for s in self.shms]

    def all_reduce_sum_(self, x, slow_reader=0.0):
        n = x.size
        assert n <= self.slot
        self.seq += 1
        par = self.seq & 1
        for p in range(self.world):
            self.data[p][par, self.rank, :n] = x
        for p in range(self.world):
            self.flags[p][par, self.rank, 0] = self.seq
        t0 = time.time()
        while any(self.flags[self.rank][par, r, 0] < self.seq for r in range(self.world)):
            if time.time() - t0 > 20:
                raise TimeoutError("rank %d waited 20 s in collective %d" % (self.rank, self.seq))
            time.sleep(0)
        if slow_reader:
            time.sleep(slow_reader)       # a peer may already be writing the NEXT collective: the other parity
        s = np.zeros(n)
        for r in range(self.world):
            s += self.data[self.rank][par, r, :n]
        x[:] = s
        return x

    def close(self, unlink=False):
        self.data = self.flags = None
        for r, s in enumerate(self.shms):
            if s is not None:
                s.close()
                if unlink and r == self.rank:
                    s.unlink()
