"""TEST-ONLY stand-in for gpu-physics-engine_amd.sharded.GpeEngine: one rank's particles on the CPU, stepped by
the oracle.  It lets the world_size-2 gloo tests exercise the sharding logic (ownership, ghost band,
migration, global re-sort indices) without a GPU.  Never imported by the product path."""
import contextlib

import numpy as np
import torch


class OracleEngine:
    device = torch.device("cpu")

    def __init__(self, orc, pos, rad, gid, world, cell_size, gravity=(0.0, 0.0), capacity=None):
        self.orc = orc
        n = len(rad)
        cap = capacity or int(n * 1.5) + 64
        self.cap = cap
        self.a = {"pos": torch.zeros((cap, 2), dtype=torch.float32), "prev": torch.zeros((cap, 2), dtype=torch.float32),
                  "radius": torch.zeros(cap, dtype=torch.float32), "gid": torch.zeros(cap, dtype=torch.int32)}
        self.a["pos"][:n] = torch.from_numpy(np.ascontiguousarray(pos, np.float32).reshape(-1, 2))
        self.a["prev"][:n] = self.a["pos"][:n]
        self.a["radius"][:n] = torch.from_numpy(np.ascontiguousarray(rad, np.float32))
        self.a["gid"][:n] = torch.from_numpy(np.ascontiguousarray(gid, np.int64)).to(torch.int32)
        self.n_owned, self.n_total = n, n
        self.params = orc.default_params(world[0], world[1], 0.5, gravity=gravity)
        self.params.cell_size = np.float32(cell_size)

    def capacity(self):
        return self.cap

    def reserve(self, capacity):
        if capacity <= self.cap:
            return
        for k, v in self.a.items():
            nv = torch.zeros((capacity,) + tuple(v.shape[1:]), dtype=v.dtype)
            nv[:self.cap] = v
            self.a[k] = nv
        self.cap = capacity

    def arrays(self):
        return self.a

    def set_counts(self, n_total, n_owned):
        self.n_total, self.n_owned = n_total, n_owned

    def make_tables(self, dec, cap):
        return None

    def classify(self, dec, rank, tables):
        pos = self.a["pos"][:self.n_owned].numpy()
        cx = np.floor(pos[:, 0] / dec.cell_size).astype(np.int64) >> 3
        cy = np.floor(pos[:, 1] / dec.cell_size).astype(np.int64) >> 3
        cx = np.clip(cx, 0, dec.bx - 1); cy = np.clip(cy, 0, dec.by - 1)
        owner = dec.owner[cy, cx].astype(np.int64)
        mask = dec.dest_mask[cy, cx].astype(np.int64)
        info = (mask & 0x03FFFFFF) | np.where(owner != rank, (owner + 1) << 26, 0)      # gpe.h gpe_shard_classify
        sel = np.nonzero(info)[0]
        return torch.from_numpy(sel), torch.from_numpy(info[sel].astype(np.int32))

    def step(self, dt):
        """Collisions on owned + ghosts in ascending order key (== the unsharded object index order), then
        Verlet on the owned particles only."""
        orc, n, no = self.orc, self.n_total, self.n_owned
        pos = self.a["pos"][:n].numpy(); rad = self.a["radius"][:n].numpy()
        order = np.argsort(self.a["gid"][:n].numpy().astype(np.int64), kind="stable")
        sim = orc.Sim(pos[order], rad[order], self.params)
        sim.grid_build(); sim.grid_sort(); sim.build_collision_cells(); sim.solve_colors()
        out = np.empty_like(pos)
        out[order] = sim.pos
        sim.close()
        npos, nprev = orc.verlet_integration(out[:no], self.a["prev"][:no].numpy(), rad[:no], self.params, dt)
        self.a["pos"][:no] = torch.from_numpy(npos)
        self.a["prev"][:no] = torch.from_numpy(nprev)

    def morton_resort(self):
        n = self.n_owned
        pos = self.a["pos"][:n].numpy()
        keys, _ = self.orc.create_home_cell_ids(pos, float(self.params.cell_size))
        perm = np.argsort(keys, kind="stable")
        for k in ("pos", "prev", "radius"):
            self.a[k][:n] = self.a[k][:n][torch.from_numpy(perm)]
        return torch.from_numpy(keys[perm].astype(np.int64)), torch.from_numpy(perm)

    def argsort_u32(self, keys):
        return torch.from_numpy(np.argsort(keys.numpy().astype(np.int64) & 0xFFFFFFFF, kind="stable"))

    def set_active_cells(self, box): pass
    def refresh(self): pass
    def sync(self): pass
    def close(self): pass

    def stream_ctx(self):
        return contextlib.nullcontext()
