"""AdamW over every parameter tensor of a group in ONE launch (csrc/heads.hip: adamw_kernel).

Same update as `torch.optim.AdamW` (decoupled weight decay, bias-corrected moments, no amsgrad / maximize), which the
reference configures through Lightning (model/pipeline.py:738-757); `torch.optim.AdamW(fused=True)` issues one
multi-tensor launch per ~30 tensors (8 launches, 0.27 ms for the detector's ~300 tensors), this one issues one.
A device table of (param, grad, exp_avg, exp_avg_sq) pointers is kept per group.  Per step the host only collects the
gradient addresses (the native U-Net executor keeps its flat gradient buffer, the few head gradients are re-allocated by
`zero_grad(set_to_none=True)`); when any of them moved, the address column is refreshed through a pinned staging buffer
(one asynchronous 10 KB copy, no synchronisation).

The step count is PER PARAMETER, as in `torch.optim.AdamW` (bias correction `1 - beta^t` with t = the number of updates this
tensor has received): ScoreNet / `score_linear` only start receiving gradients once proposals exist
(`epoch > prepare_epochs`, model/pointgroup.py:332) and must then start at t = 1, not at the backbone's t.  The tensors of a
table are ordered by their step count, so tensors with equal counts ("cohorts") own a contiguous range of the block map and
get one launch with their own bias corrections -- one launch in the steady state.  `state_dict()` carries
`state[p]["step"]`, so checkpoints move to and from `torch.optim.AdamW` in both directions."""
import math
import operator

import numpy as np
import torch

from . import _lib
from ._lib import check


_DATA_PTR = operator.methodcaller("data_ptr")
_GRAD = operator.attrgetter("grad")


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}

    RING = 4   # pinned staging slots per group: an address refresh never rewrites a buffer whose copy may still be queued

    def _flush_steps(self):
        """cohort step counters of the cached tables -> state[p]["step"] (python ints)"""
        for tb in self._tables.values():
            for t, lo, hi, _b0, _nb in tb.get("cohorts", ()):
                for p in tb["plist"][lo:hi]:
                    self.state[p]["step"] = t

    def state_dict(self):
        """torch.optim.AdamW's layout: per-parameter `step` (a float32 scalar tensor), `exp_avg`, `exp_avg_sq`.  The returned
        per-parameter dicts are COPIES: `Optimizer.state_dict()` hands out the live `self.state[p]` objects, and this class keeps
        python-int step counts there (converting in place also rewrote the dict already returned)."""
        self._flush_steps()
        sd = super().state_dict()
        sd["state"] = {k: (dict(st, step=torch.tensor(float(st["step"]))) if "step" in st and not torch.is_tensor(st["step"]) else dict(st))
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        """moments are replaced: the cached device tables (which hold their addresses) are dropped; the step counts come from
        the checkpoint: torch.optim.AdamW's per-parameter `state[p]["step"]`, or -- checkpoints written by the first
        version of this class -- one per-group `step` applied to every tensor that has moments"""
        super().load_state_dict(state_dict)
        self._tables = {}
        for group in self.param_groups:
            legacy = group.pop("step", None)
            for p in group["params"]:
                st = self.state.get(p)
                if st is None or "exp_avg" not in st:
                    continue
                if "step" in st:
                    st["step"] = int(st["step"])
                elif legacy is not None:
                    st["step"] = int(legacy)

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}

    @staticmethod
    def _validate(p):
        g = p.grad
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()
                and g.dtype == torch.float32 and not g.is_sparse and g.device == p.device):
            raise RuntimeError("FusedAdamW: contiguous fp32 device parameters and gradients only")

    def _build(self, gi, plist):
        for p in plist:
            self._validate(p)
        # equal step counts adjacent (most-stepped first; stable): one contiguous block-map range, one launch per cohort
        plist.sort(key=lambda p: -int(self.state[p].get("step", 0)))
        dev = plist[0].device
        fresh = [p for p in plist if "exp_avg" not in self.state[p]]
        if fresh:   # moments of the tensors seen for the first time: one flat buffer
            flat = torch.zeros(2 * sum(p.numel() for p in fresh), dtype=torch.float32, device=dev)
            o = 0
            for p in fresh:
                n = p.numel()
                self.state[p]["exp_avg"] = flat[o:o + n].view_as(p)
                self.state[p]["exp_avg_sq"] = flat[o + n:o + 2 * n].view_as(p)
                o += 2 * n
        chunk = _lib.lib().d3_adamw_chunk()
        ring = [torch.empty((len(plist), 4), dtype=torch.int64).pin_memory() for _ in range(self.RING)]
        host = ring[0]
        hv = host.numpy()
        hv[:, 0] = [p.data_ptr() for p in plist]
        hv[:, 1] = [p.grad.data_ptr() for p in plist]
        hv[:, 2] = [self.state[p]["exp_avg"].data_ptr() for p in plist]
        hv[:, 3] = [self.state[p]["exp_avg_sq"].data_ptr() for p in plist]
        numel = np.array([p.numel() for p in plist], dtype=np.int32)
        blocks = np.array([(t, c) for t, n in enumerate(numel) for c in range((int(n) + chunk - 1) // chunk)], dtype=np.int32)
        nblk = [(int(n) + chunk - 1) // chunk for n in numel]
        cohorts, lo, b0 = [], 0, 0      # [step count, first tensor, end tensor, first block, blocks]
        for i in range(1, len(plist) + 1):
            if i == len(plist) or int(self.state[plist[i]].get("step", 0)) != int(self.state[plist[lo]].get("step", 0)):
                nb = sum(nblk[lo:i])
                cohorts.append([int(self.state[plist[lo]].get("step", 0)), lo, i, b0, nb])
                lo, b0 = i, b0 + nb
        tb = {"pptr": [p.data_ptr() for p in plist], "gptr": hv[:, 1].tolist(), "host": host, "ring": ring, "slot": 0, "cohorts": cohorts,
              "mptr": [self.state[p]["exp_avg"].data_ptr() for p in plist],
              "nblocks": int(blocks.shape[0]), "ptrs": host.to(dev), "numel": torch.from_numpy(numel).to(dev),
              "blocks": torch.from_numpy(blocks.reshape(-1)).to(dev), "device": dev}
        self._tables[gi] = tb
        return tb

    def _table(self, gi, group):
        tb = self._tables.get(gi)
        params = group["params"]
        if tb is not None and len(tb["plist"]) <= len(params):
            plist = tb["plist"]
            try:
                gptr = list(map(_DATA_PTR, map(_GRAD, plist)))
            except AttributeError:          # a gradient went missing: rebuild over the tensors that have one
                gptr = None
            if gptr is not None and tb["ngrad"] == sum(1 for p in params if p.grad is not None) \
                    and tb["pptr"] == list(map(_DATA_PTR, plist)):
                if gptr != tb["gptr"]:   # some gradients were re-allocated: refresh their addresses (asynchronous, stream ordered)
                    for i, (a, b) in enumerate(zip(gptr, tb["gptr"])):
                        if a != b:
                            self._validate(plist[i])
                    # next pinned slot of the ring: the previous refresh's host->device copy may still be in flight
                    tb["slot"] = (tb["slot"] + 1) % self.RING
                    nxt = tb["ring"][tb["slot"]]
                    nxt.copy_(tb["host"]); tb["host"] = nxt
                    tb["host"].numpy()[:, 1] = gptr
                    tb["ptrs"].copy_(tb["host"], non_blocking=True)
                    tb["gptr"] = gptr
                return tb
        plist = [p for p in params if p.grad is not None]
        if not plist:
            return {"nblocks": 0, "plist": [], "ngrad": -1}
        self._flush_steps()            # the table being replaced owns the current counts
        self._tables.pop(gi, None)
        tb = self._build(gi, plist)
        tb["plist"] = plist
        tb["ngrad"] = len(plist)
        return tb

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            tb = self._table(gi, group)
            if tb["nblocks"] == 0:
                continue
            b1, b2 = group["betas"]
            with torch.cuda.device(tb["device"]):
                st = torch.cuda.current_stream().cuda_stream
                for co in tb["cohorts"]:       # tensors with the same number of updates behind them: one launch
                    co[0] = t = co[0] + 1
                    check(L.d3_adamw(tb["ptrs"].data_ptr(), tb["numel"].data_ptr(), tb["blocks"].data_ptr() + 8 * co[3], co[4],
                                     float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                     1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), st), "adamw")
        return loss
