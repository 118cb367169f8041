"""AdamW over every parameter tensor of a group in ONE launch (csrc/heads.hip: adamw_kernel).

Same update as `torch.optim.AdamW` (decoupled weight decay, bias-corrected moments, no amsgrad / maximize), which the
reference configures through Lightning (model/pipeline.py:738-757); `torch.optim.AdamW(fused=True)` issues one
multi-tensor launch per ~30 tensors (8 launches, 0.27 ms for the detector's ~300 tensors), this one issues one.
A device table of (param, grad, exp_avg, exp_avg_sq) pointers is kept per group and rebuilt only when a gradient tensor
was replaced (the native U-Net executor keeps its flat gradient buffer, so in steady state it never is)."""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}

    def _table(self, gi, group):
        plist = [p for p in group["params"] if p.grad is not None]
        grads = [p.grad for p in plist]
        tb = self._tables.get(gi)
        if tb is not None and len(tb["grads"]) == len(grads) and all(a is b for a, b in zip(tb["grads"], grads)) \
                and all(a is b for a, b in zip(tb["params"], plist)):
            return tb
        for p in plist:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                    and p.grad.dtype == torch.float32 and not p.grad.is_sparse):
                raise RuntimeError("FusedAdamW: contiguous fp32 device parameters and gradients only")
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
        ptrs = np.array([[p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                          self.state[p]["exp_avg_sq"].data_ptr()] for p in plist], dtype=np.int64)
        numel = np.array([p.numel() for p in plist], dtype=np.int32)
        blocks = np.array([(t, c) for t, n in enumerate(numel) for c in range((int(n) + chunk - 1) // chunk)], dtype=np.int32)
        tb = {"grads": grads, "params": plist, "nblocks": int(blocks.shape[0]),
              "ptrs": torch.from_numpy(ptrs).to(dev), "numel": torch.from_numpy(numel).to(dev),
              "blocks": torch.from_numpy(blocks.reshape(-1)).to(dev), "device": dev}
        self._tables[gi] = tb
        return tb

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            if not any(p.grad is not None for p in group["params"]):
                continue
            tb = self._table(gi, group)
            group["step"] = t = int(group.get("step", 0)) + 1
            b1, b2 = group["betas"]
            with torch.cuda.device(tb["device"]):
                check(L.d3_adamw(tb["ptrs"].data_ptr(), tb["numel"].data_ptr(), tb["blocks"].data_ptr(), tb["nblocks"],
                                 float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t),
                                 torch.cuda.current_stream().cuda_stream), "adamw")
        return loss
