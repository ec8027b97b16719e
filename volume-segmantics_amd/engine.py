"""VolSegUnet: the object ``create_model_on_device`` returns for U_NET + resnet34.

It looks like the ``smp.Unet`` the reference builds (volume_segmantics/model/model_2d.py:15-16):
``__call__((B,1,H,W) fp32) -> (B,K,H,W) fp32 logits``, ``train()/eval()``, ``named_parameters()``
with smp's names (so the reference's freeze predicate works, vol_seg_2d_trainer.py:102-116),
``state_dict()/load_state_dict()`` with smp's keys and OIHW shapes, autograd-compatible output so
``loss.backward()`` populates ``.grad`` (vol_seg_2d_trainer.py:429-430).

Underneath there is no torch arithmetic: all parameters are views of ONE flat fp32 buffer (conv
weights stored [cout][kh][kw][cin] = torch channels_last), and forward / backward are single calls
into libvolseg_hip.so (vs_unet_forward / vs_unet_backward) on the current HIP stream.  PyTorch
tensors are only the memory container.
"""
from __future__ import annotations

import logging
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, lib, ptr

KIND_CONV, KIND_GAMMA, KIND_BETA, KIND_BIAS, KIND_RMEAN, KIND_RVAR = range(6)


def default_precision() -> str:
    return os.environ.get("VOLSEG_PRECISION", "fp32")


class _Node(nn.Module):
    """Name-space container so that parameter names come out as smp's dotted keys."""


def _attach(root: nn.Module, dotted: str, tensor, is_param: bool):
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    if is_param:
        node.register_parameter(parts[-1], tensor)
    else:
        node.register_buffer(parts[-1], tensor)


class _UnetFn(torch.autograd.Function):
    """One autograd node for the whole network.  Weight gradients are written by the library
    into the model's flat gradient buffer and attached to ``param.grad`` directly (views, no
    per-tensor copies); the returned gradients for the parameter inputs are therefore None."""

    @staticmethod
    def forward(ctx, x, model, anchor):
        ctx.model = model
        ctx.x = x
        ctx.token = model._train_forward_token + 1
        return model._forward_impl(x, training=True)

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if ctx.token != model._train_forward_token:
            raise RuntimeError("VolSegUnet: backward() must follow the training forward that produced this output "
                               "(activations live in a single workspace; only the latest forward can be differentiated)")
        model._backward_impl(ctx.x, dlogits)
        return None, None, None


class VolSegUnet(nn.Module):
    ENCODERS = {"resnet18": 18, "resnet34": 34, "resnet50": 50, "resnext50_32x4d": 51,
                "efficientnet-b3": 103, "efficientnet-b4": 104,      # every topology but Linknet (csrc/unet.hip build())
                "timm-resnest50d": 150, "timm-resnest101e": 201}     # not under the dilating decoders (DeepLabV3(+), PAN)
    # efficientnet-pytorch registers these and smp's encoder never runs them: torch leaves their .grad at None, AdamW skips them
    UNUSED_PREFIXES = ("encoder._conv_head.", "encoder._bn1.")
    TOPOLOGIES = {"unet": 0, "unetplusplus": 1, "linknet": 2, "fpn": 3,
                  "deeplabv3plus": 4, "deeplabv3": 5, "manet": 6,
                  "pan": 7}     # smp.Unet, UnetPlusPlus, Linknet, FPN, DeepLabV3Plus, DeepLabV3, MAnet, PAN

    def __init__(self, classes: int, device=None, precision: str | None = None, init: str = "smp", seed: int | None = None,
                 encoder: str = "resnet34", topology: str = "unet"):
        super().__init__()
        if encoder not in self.ENCODERS:
            raise NotImplementedError(f"encoder {encoder!r}: the engine builds {sorted(self.ENCODERS)}")
        if topology not in self.TOPOLOGIES:
            raise NotImplementedError(f"topology {topology!r}: the engine builds {sorted(self.TOPOLOGIES)}")
        self.encoder_name, self.topology = encoder, topology
        self._enc = self.TOPOLOGIES[topology] * 1000 + self.ENCODERS[encoder]     # the C ABI's encoder code
        precision = precision or default_precision()
        if precision not in ("fp32", "bf16", "fp16"):
            raise ValueError(f"precision must be 'fp32', 'bf16' or 'fp16' (inference only), got {precision!r}")
        self.classes = int(classes)
        self.precision = precision
        self._dtype_code = {"fp32": _lib.VS_F32, "bf16": _lib.VS_BF16, "fp16": _lib.VS_F16}[precision]
        self._table = _lib.unet_tensor_table(self.classes, self._enc)
        dev = torch.device(device if device is not None else "cpu")
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        n_params = lib.vs_unet_param_elems_ex(self.classes, self._enc)
        n_bn = lib.vs_unet_bnstate_elems_ex(self.classes, self._enc)
        self._flat = torch.zeros(n_params, dtype=torch.float32, device=dev)
        self._flat_grad = None
        self._bnstate = torch.zeros(n_bn, dtype=torch.float32, device=dev)
        n_bns = sum(1 for t in self._table if t[2] == KIND_RMEAN)
        self._nbt = torch.zeros(n_bns, dtype=torch.int64, device=dev)
        self._views: dict[str, torch.Tensor] = {}
        self._build_views()
        self._plans: dict = {}
        self._prep_key = None
        self._train_forward_token = 0
        self._anchor = None
        self.dp_group = None  # torch.distributed process group for data-parallel gradient all-reduce
        self.dp_grad_dtype = torch.float32
        self.dp_buckets = 4  # >1: bucketed, overlapped gradient all-reduce (decoder+head, layer4, layer3, rest)
        self.sync_bn = False  # data parallel: BatchNorm statistics of the GLOBAL batch (SyncBatchNorm; see _sync_hook) instead of per rank
        self._wver = 0   # bumped when a HIP kernel (not a torch op) rewrites the parameters
        self._fused_optimizer = None   # FusedAdamW(fuse_step_into_backward=True) registers itself here
        self.dropout_seed = 0 if seed is None else int(seed)   # Dropout2d draws (smp.FPN); give every data-parallel rank its own
        self._dp_side = None           # side stream of the data-parallel fused optimiser step
        self._bnver = 0  # bumped when a training forward moves the running statistics
        self._step_side = None         # second stream of a replayed step (weight gradients + optimiser)
        self._steps: dict = {}         # captured training steps (fused_train_step): static buffers + one hipGraph per weight set
        if init == "smp":
            self.reset_parameters(seed)

    # ------------------------------------------------------------------ parameter plumbing
    def _view_of(self, flat, shape, kind, off):
        numel = math.prod(shape)
        v = flat[off:off + numel]
        if kind == KIND_CONV:
            o, i, kh, kw = shape
            return v.view(o, kh, kw, i).permute(0, 3, 1, 2)  # OIHW shape, KRSC memory
        return v.view(shape)

    def _build_views(self):
        for name in list(self._modules):
            del self._modules[name]
        bn_i = 0
        for name, shape, kind, off in self._table:
            if kind <= KIND_BIAS:
                p = nn.Parameter(self._view_of(self._flat, shape, kind, off), requires_grad=True)
                _attach(self, name, p, True)
                self._views[name] = p
            else:
                b = self._view_of(self._bnstate, shape, kind, off)
                _attach(self, name, b, False)
                self._views[name] = b
                if kind == KIND_RVAR:
                    _attach(self, name.rsplit(".", 1)[0] + ".num_batches_tracked", self._nbt[bn_i], False)
                    bn_i += 1
        self._param_cache = None
        self._unused_ids = {id(self._views[t[0]]) for t in self._table if t[2] <= KIND_BIAS and t[0].startswith(self.UNUSED_PREFIXES)}

    def reset_parameters(self, seed: int | None = None):
        """smp / torchvision initialisation: encoder convs kaiming_normal(fan_out, relu),
        decoder convs kaiming_uniform(fan_in, relu), head xavier_uniform, biases 0, BN weight 1 / bias 0,
        running_mean 0 / running_var 1."""
        gen = None
        if seed is not None:
            gen = torch.Generator(device="cpu").manual_seed(seed)
        convt_bound = None
        with torch.no_grad():
            effnet = self.encoder_name.startswith("efficientnet")
            fan_in = None
            for name, shape, kind, off in self._table:
                v = self._views[name]
                if kind == KIND_CONV and effnet and name.startswith("encoder."):
                    # efficientnet-pytorch initialises nothing itself: nn.Conv2d's default (kaiming_uniform(a = sqrt(5)), uniform bias)
                    w = torch.empty(shape, dtype=torch.float32)
                    nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=gen)
                    v.copy_(w)
                    fan_in = shape[1] * shape[2] * shape[3]
                elif kind == KIND_BIAS and effnet and name.startswith("encoder.") and len(shape) == 1:
                    bound = 1.0 / math.sqrt(fan_in)
                    v.copy_(torch.empty(shape, dtype=torch.float32).uniform_(-bound, bound, generator=gen))
                elif kind == KIND_CONV:
                    w = torch.empty(shape, dtype=torch.float32)
                    if name.startswith("encoder."):
                        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu", generator=gen)
                    elif name.startswith("decoder."):
                        nn.init.kaiming_uniform_(w, mode="fan_in", nonlinearity="relu", generator=gen)
                    else:
                        nn.init.xavier_uniform_(w, generator=gen)
                    v.copy_(w)
                elif kind in (KIND_GAMMA, KIND_RVAR):
                    v.fill_(1.0)
                elif kind == KIND_BIAS and len(shape) == 4:
                    # nn.ConvTranspose2d (Linknet's TransposeX2) is not an nn.Conv2d: smp's initialize_decoder leaves torch's
                    # default initialisation in place - kaiming_uniform(a = sqrt(5)) and a uniform bias (the tensor after it)
                    w = torch.empty(shape, dtype=torch.float32)
                    nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=gen)
                    v.copy_(w)
                    convt_bound = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])
                elif kind == KIND_BIAS and convt_bound is not None:
                    v.copy_(torch.empty(shape, dtype=torch.float32).uniform_(-convt_bound, convt_bound, generator=gen))
                    convt_bound = None
                else:
                    v.zero_()
            self._nbt.zero_()

    def _apply(self, fn, recurse=True):
        """.to(device) / .cuda() / .float(): move the flat buffers and rebuild the views."""
        new_flat = fn(self._flat)
        if new_flat.dtype != torch.float32:
            raise TypeError("VolSegUnet keeps fp32 master parameters; choose the compute precision with precision=")
        moved = new_flat.device != self._flat.device
        if moved or new_flat is not self._flat:
            req = {n: p.requires_grad for n, p in self.named_parameters()}
            self._flat = new_flat.contiguous()
            self._bnstate = fn(self._bnstate).contiguous()
            self._nbt = fn(self._nbt).contiguous()
            self._flat_grad = None
            self._build_views()
            for n, p in self.named_parameters():
                p.requires_grad = req[n]
            self._drop_steps()
            self._plans.clear()
            self._prep_key = None
        return self

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        out = OrderedDict()
        for k, v in sd.items():  # plain contiguous OIHW tensors: interchangeable with smp checkpoints
            out[k] = v.detach().clone(memory_format=torch.contiguous_format)
        if hasattr(sd, "_metadata"):
            out._metadata = sd._metadata
        return out

    @property
    def device(self):
        return self._flat.device

    # ------------------------------------------------------------------ plans / workspace
    def _plan(self, n: int, h: int, w: int, training: bool):
        if self.device.type != "cuda":
            raise RuntimeError("VolSegUnet: the HIP engine needs a GPU device; there is no CPU fallback "
                               "(the CPU oracle lives under oracle/ and is test infrastructure only)")
        if h % 32 or w % 32:
            raise ValueError(f"VolSegUnet: input height/width must be multiples of 32, got {h}x{w}")
        key = (h, w)
        plan = self._plans.get(key)
        if plan is None or plan["max_batch"] < n or (training and not plan["training"]):
            if plan is not None:
                lib.vs_unet_destroy(plan["handle"])
            handle = _lib.C.c_void_p()
            max_batch = max(n, plan["max_batch"] if plan else 0)
            check(lib.vs_unet_create_ex(_lib.C.byref(handle), self._dtype_code, self.classes, max_batch, h, w, self._enc))
            train_ws = training or (plan is not None and plan["training"])
            nbytes = lib.vs_unet_workspace_bytes(handle, 1 if train_ws else 0)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            plan = {"handle": handle, "max_batch": max_batch, "training": train_ws, "ws": ws, "prep": None, "rng": None}
            self._plans[key] = plan
        if training:
            self._sync_hook(plan)
        if training and plan["rng"] != (self.dropout_seed, ptr(self._nbt)):
            # Dropout2d draws (smp.FPN): mask = f(seed, encoder.bn1.num_batches_tracked) - the counter every training step advances
            check(lib.vs_unet_set_rng(plan["handle"], self.dropout_seed & 0xFFFFFFFF, ptr(self._nbt)))
            plan["rng"] = (self.dropout_seed, ptr(self._nbt))
        return plan

    def _sync_hook(self, plan):
        """SyncBatchNorm (``model.sync_bn = True`` with a data-parallel group of more than one rank): the library calls back after every
        convolution that feeds a BatchNorm, and in every BatchNorm backward, with a slice of the plan's workspace to be summed over
        the ranks - 64-bit fixed-point statistics sums in the forward pass (integer adds: every rank ends with the same bits, and
        with the bits a single process running the whole global batch would form), two fp32 vectors in the backward pass.  Every rank
        must run the same batch size.  The reference has one loader and one BatchNorm batch (data/dataloaders.py:42-49); this is
        what lets N ranks reproduce it.  Recorded steps (hipGraphs) cannot carry the callback: can_fuse_step is False."""
        want = bool(self.sync_bn) and self.dp_group is not None and self._world() > 1
        have = plan.get("sync")
        if not want:
            if have is not None:
                check(lib.vs_unet_set_stats_hook(plan["handle"], None, None, 1))
                plan["sync"] = None
            return
        key = (id(self.dp_group), self._world(), plan["ws"].data_ptr())
        if have is not None and have[0] == key:
            return
        ws, group = plan["ws"], self.dp_group
        base, nbytes = ws.data_ptr(), ws.numel()
        state = {"error": None}

        def hook(_user, values, count, kind, _stream):
            try:
                off = int(values) - base
                width = 8 if kind == 0 else 4
                if off < 0 or off + count * width > nbytes or off % width:
                    raise RuntimeError("statistics hook called with a buffer outside the plan's workspace")
                t = ws[off:off + count * width].view(torch.int64 if kind == 0 else torch.float32)
                import torch.distributed as dist
                dist.all_reduce(t, group=group)      # stream-ordered on the current stream (nccl = RCCL), synchronous under gloo
                return 0
            except Exception as e:  # noqa: BLE001 - an exception must not unwind through the C frames
                state["error"] = e
                return -1

        cb = _lib.STATS_HOOK(hook)
        check(lib.vs_unet_set_stats_hook(plan["handle"], _lib.C.cast(cb, _lib.C.c_void_p), None, self._world()))
        plan["sync"] = (key, cb, state)      # (the CFUNCTYPE object must outlive the registration)

    def _drop_steps(self):
        for st in self._steps.values():
            for prog in st["graphs"].values():
                for op, obj in (prog or []):
                    if op in ("main", "side"):
                        lib.vs_graph_destroy(obj)
        self._steps.clear()

    def release_plans(self):
        """Free every plan (and the step graphs recorded against them); they are rebuilt on demand."""
        self._drop_steps()
        for plan in self._plans.values():
            lib.vs_unet_destroy(plan["handle"])
        self._plans.clear()

    def __del__(self):
        try:
            self.release_plans()
        except Exception:
            pass

    def _prepare(self, plan, training: bool):
        key = (self._flat._version, self._wver, training,
               None if training else (self._bnstate._version, self._bnver))
        if plan["prep"] != key:
            check(lib.vs_unet_prepare(plan["handle"], ptr(self._flat), ptr(self._bnstate), 1 if training else 0,
                                      ptr(plan["ws"]), _lib.stream_ptr()))
            plan["prep"] = key

    # ------------------------------------------------------------------ forward / backward
    def _check_input(self, x):
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"VolSegUnet expects (B,1,H,W) input, got {tuple(x.shape)}")
        if x.device != self.device:
            raise RuntimeError(f"input on {x.device}, model on {self.device}")
        if x.dtype != torch.float32:
            x = x.float()
        return x.contiguous()

    def _forward_impl(self, x, training: bool):
        n, _, h, w = x.shape
        plan = self._plan(n, h, w, training)
        self._prepare(plan, training)
        logits = torch.empty((n, self.classes, h, w), dtype=torch.float32, device=self.device)
        if training:
            self._nbt += 1      # before the forward: the dropout draw of this step reads the advanced counter, as a replayed step does
        check(lib.vs_unet_forward(plan["handle"], ptr(self._flat), ptr(self._bnstate), ptr(x), n, 1 if training else 0,
                                  ptr(logits), ptr(plan["ws"]), _lib.stream_ptr()))
        if training:
            self._train_forward_token += 1
            self._bnver += 1
        return logits

    def _forward_to_volume(self, x, dmap, s0, mode, direction, labels, probs, keys, votes, nvox):
        """Prediction batch: eval forward of the padded slices x whose head lands in the output volume(s) (predictor)."""
        n, _, h, w = x.shape
        plan = self._plan(n, h, w, False)
        self._prepare(plan, False)
        check(lib.vs_unet_forward_to_volume(plan["handle"], ptr(self._flat), ptr(self._bnstate), ptr(x), n, ptr(plan["ws"]),
                                            _lib.stream_ptr(), dmap, s0, mode, direction, ptr(labels), ptr(probs), ptr(keys),
                                            ptr(votes), nvox))

    def _backward_impl(self, x, dlogits):
        n, _, h, w = x.shape
        plan = self._plans[(h, w)]
        if self._flat_grad is None:
            self._flat_grad = torch.zeros_like(self._flat)
        if self._param_cache is None:
            named = dict(self.named_parameters())
            self._param_cache = [(named[t[0]], t[1], t[2], t[3], "encoder" in t[0] and "conv" in t[0])
                                 for t in self._table if t[2] <= KIND_BIAS]
        need_enc = any(p.requires_grad for p, _, _, _, enc in self._param_cache if enc)
        dlogits = dlogits.contiguous()
        if dlogits.dtype != torch.float32:
            dlogits = dlogits.float()
        fused = self._fused_optimizer
        if fused is not None and not fused._can_fuse(self, need_enc):
            fused = None
        if fused is not None and self.dp_group is not None and self._world() > 1:
            # data parallel: every bucket is all-reduced, then its AdamW step and weight copies run on a side stream while the
            # backward of the layers below it continues
            self._backward_bucketed(plan, x, dlogits, n, need_enc, fused=fused)
            fused._stepped_in_backward = True
            self._wver += 1
            plan["prep"] = (self._flat._version, self._wver, True, None)
        elif fused is not None:
            # the optimiser step rides on the backward's side stream (FusedAdamW(fuse_step_into_backward=True))
            g = fused.param_groups[0]
            args = _lib.AdamwArgs(ptr(self._flat), ptr(fused.exp_avg), ptr(fused.exp_avg_sq), float(g["lr"]),
                                  float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                                  fused.step_count + 1)
            check(lib.vs_unet_backward_adamw(plan["handle"], ptr(x), ptr(dlogits), n, 1 if need_enc else 0,
                                             ptr(self._flat_grad), ptr(plan["ws"]), _lib.stream_ptr(), _lib.C.byref(args)))
            fused._stepped_in_backward = True
            self._wver += 1
            plan["prep"] = (self._flat._version, self._wver, True, None)   # the next forward's copies are already in place
        elif self.dp_group is not None and self.dp_buckets > 1 and self._world() > 1:
            self._backward_bucketed(plan, x, dlogits, n, need_enc)
        else:
            check(lib.vs_unet_backward(plan["handle"], ptr(self._flat), ptr(x), ptr(dlogits), n, 1 if need_enc else 0,
                                       ptr(self._flat_grad), ptr(plan["ws"]), _lib.stream_ptr()))
            if self.dp_group is not None:
                self._allreduce_grads()
        self._attach_grads()

    def _attach_grads(self, accumulate: bool = True):
        for p, shape, kind, off, _ in self._param_cache:
            if not p.requires_grad or id(p) in self._unused_ids:
                continue
            g = self._view_of(self._flat_grad, shape, kind, off)
            if p.grad is None:
                p.grad = g
            elif p.grad.data_ptr() != g.data_ptr():
                if accumulate:
                    p.grad.add_(g)
                else:
                    p.grad = g
            # else: .grad already aliases the flat buffer, which now holds this step's gradient

    # ------------------------------------------------------------------ captured training step
    def _ensure_param_cache(self):
        if self._flat_grad is None:
            self._flat_grad = torch.zeros_like(self._flat)
        if self._param_cache is None:
            named = dict(self.named_parameters())
            self._param_cache = [(named[t[0]], t[1], t[2], t[3], "encoder" in t[0] and "conv" in t[0])
                                 for t in self._table if t[2] <= KIND_BIAS]
        return any(p.requires_grad for p, _, _, _, enc in self._param_cache if enc)

    def can_fuse_step(self, opt, x, targets) -> bool:
        """Whether fused_train_step applies: single process, the model's own fused AdamW, Dice targets of the logits' shape,
        "everything trains" or "everything but the encoder convolutions" (the reference's two phases)."""
        if os.environ.get("VOLSEG_STEP_GRAPH", "1") == "0" or not self.training or self.device.type != "cuda":
            return False
        if not isinstance(opt, FusedAdamW) or opt.model is not self or self._fused_optimizer is not opt:
            return False
        if self.dp_group is not None and self._world() > 1 and (self.dp_grad_dtype != torch.float32 or self.sync_bn):
            return False
        if x.dim() != 4 or targets.dim() != 4 or targets.shape != (x.shape[0], self.classes, x.shape[2], x.shape[3]):
            return False
        if targets.dtype not in (torch.uint8, torch.float32) or lib.vs_profile_enabled():
            return False
        need_enc = self._ensure_param_cache()
        return all(p.requires_grad == (need_enc or not enc) for p, _, _, _, enc in self._param_cache)

    def fused_train_step(self, x, targets, opt: "FusedAdamW", eps: float = 1e-6, clone_loss: bool = True):
        """The reference's whole ``_train_one_batch`` (vol_seg_2d_trainer.py:419-432) - zero_grad, forward,
        DiceLoss(normalization="none"), backward, AdamW step (the scheduler's current lr / beta1) - as ONE replayed hipGraph:
        the ~430 launches of a step are recorded once per weight set (vs_capture_begin .. vs_capture_end) with the optimiser's
        scalars in device memory, and every later step costs the host two calls.  Same kernels in the same order as
        ``model(x)`` -> ``HipDiceLoss`` -> ``loss.backward()`` -> ``opt.step()``: bit-identical parameters, optimiser state
        and running statistics (tests/test_hip_step_graph.py).  Call ``can_fuse_step`` first.  Returns the loss (0-dim device
        tensor; with ``clone_loss=False`` the step's own buffer, overwritten by the next step)."""
        x = self._check_input(x)
        n, _, h, w = x.shape
        k, hw = self.classes, h * w
        need_enc = self._ensure_param_cache()
        if opt._stepped_in_backward:
            raise RuntimeError("fused_train_step: a backward() whose optimiser step is still pending precedes this call")
        plan = self._plan(n, h, w, True)
        self._prepare(plan, True)
        is_f32 = targets.dtype == torch.float32
        key = (n, h, w, need_enc, is_f32, id(opt), opt.exp_avg.data_ptr())
        st = self._steps.get(key)
        if st is not None and st["plan"] is not plan:      # the plan was rebuilt (larger batch): recorded pointers are stale
            self._drop_steps()
            st = None
        if st is None:
            dev = self.device
            st = {"plan": plan, "x": torch.empty_like(x), "t": torch.empty_like(targets, memory_format=torch.contiguous_format),
                  "logits": torch.empty((n, k, h, w), dtype=torch.float32, device=dev),
                  "dlogits": torch.empty((n, k, h, w), dtype=torch.float32, device=dev),
                  "loss": torch.zeros((), dtype=torch.float32, device=dev),
                  "dice_ws": torch.empty(lib.vs_dice_workspace(k) // 4, dtype=torch.float32, device=dev),
                  "hyper": torch.zeros(8, dtype=torch.float32, device=dev), "graphs": {0: None, 1: None}, "eager_done": False}
            self._steps[key] = st
        if x.data_ptr() != st["x"].data_ptr():
            st["x"].copy_(x)
        if targets.data_ptr() != st["t"].data_ptr():
            st["t"].copy_(targets)
        g = opt.param_groups[0]
        stream = _lib.stream_ptr()
        check(lib.vs_train_hyper_set(ptr(st["hyper"]), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                     float(g["weight_decay"]), opt.step_count + 1, ptr(self._nbt), self._nbt.numel(), stream))

        hd = plan["handle"]
        args = _lib.AdamwArgs(ptr(self._flat), ptr(opt.exp_avg), ptr(opt.exp_avg_sq), 0.0, 0.0, 0.0, 0.0, 0.0, 1, ptr(st["hyper"]))

        def enqueue_head(s):          # forward + loss + dloss/dlogits
            check(lib.vs_unet_forward(hd, ptr(self._flat), ptr(self._bnstate), ptr(st["x"]), n, 1, ptr(st["logits"]),
                                      ptr(plan["ws"]), s))
            check(lib.vs_dice_loss_fwd(ptr(st["logits"]), ptr(st["t"]), int(is_f32), n, k, hw, eps, ptr(st["loss"]),
                                       ptr(st["dice_ws"]), st["dice_ws"].numel() * 4, s))
            check(lib.vs_dice_loss_bwd(ptr(st["logits"]), ptr(st["t"]), int(is_f32), None, n, k, hw, eps, ptr(st["dice_ws"]),
                                       ptr(st["dlogits"]), s))

        def enqueue_backward(s):      # both streams' work with fork / join events inside (flips the weight set)
            check(lib.vs_unet_backward_adamw(hd, ptr(st["x"]), ptr(st["dlogits"]), n, 1 if need_enc else 0, ptr(self._flat_grad),
                                             ptr(plan["ws"]), s, _lib.C.byref(args)))

        def enqueue_part(s, lo, hi, role):
            check(lib.vs_unet_backward_adamw_part(hd, ptr(st["x"]), ptr(st["dlogits"]), n, 1 if need_enc else 0,
                                                  ptr(self._flat_grad), ptr(plan["ws"]), s, _lib.C.byref(args), lo, hi, role))

        def enqueue_dp_part(s, lo, hi, role):      # data parallel: the two shares without the optimiser ...
            check(lib.vs_unet_backward_part(hd, ptr(self._flat), ptr(st["x"]), ptr(st["dlogits"]), n, 1 if need_enc else 0,
                                            ptr(self._flat_grad), ptr(plan["ws"]), s, lo, hi, role))

        def enqueue_dp_opt(s, lo, hi):             # ... and a bucket's AdamW + weight copies, after its all-reduce
            check(lib.vs_unet_adamw_range(hd, 1 if need_enc else 0, ptr(self._flat_grad), ptr(plan["ws"]), s,
                                          _lib.C.byref(args), lo, hi))

        def record(fn):
            cs = lib.vs_capture_begin()
            if not cs:
                raise RuntimeError(f"libvolseg_hip: {_lib.last_error()}")
            try:
                fn(cs)              # recorded, not executed
            except Exception:
                lib.vs_capture_abort()
                raise
            gh = _lib.C.c_void_p()
            check(lib.vs_capture_end(_lib.C.byref(gh)))
            return gh

        mode = os.environ.get("VOLSEG_STEP_GRAPH", "seg")
        dp = self.dp_group is not None and self._world() > 1
        if not st["eager_done"]:
            # the first step runs eagerly: lazy one-time work (side streams, events, kernel attributes) must not be recorded
            enqueue_head(stream)
            if dp:
                self._backward_bucketed(plan, st["x"], st["dlogits"], n, need_enc, fused=opt)
            else:
                enqueue_backward(stream)
            st["eager_done"] = True
        else:
            wset = lib.vs_unet_weight_set(hd)
            prog = st["graphs"][wset]
            if prog is None:
                nu = lib.vs_unet_num_units(hd)
                seg = max(1, int(os.environ.get("VOLSEG_STEP_SEG", "4")))
                if dp:
                    # data parallel: the same linear graphs without the optimiser; at the end of every gradient bucket
                    # (decoder + head, layer4, layer3, the rest) its slice is all-reduced behind the weight-gradient stream
                    # and a third stream runs the bucket's AdamW + weight copies when the all-reduce has landed
                    if "buckets" not in plan:
                        plan["buckets"] = self._bucket_plan(hd)
                    prog, first = [], True
                    for blo, bhi, a, b in plan["buckets"]:
                        cuts = list(range(bhi, blo, -seg)) + [blo]
                        for hi_, lo_ in zip(cuts[:-1], cuts[1:]):
                            if first:
                                prog.append(("main", record(lambda cs: (enqueue_head(cs), enqueue_dp_part(cs, lo_, hi_, 1)))))
                                first = False
                            else:
                                prog.append(("main", record(lambda cs: enqueue_dp_part(cs, lo_, hi_, 1))))
                            prog.append(("fork", torch.cuda.Event()))
                            prog.append(("side", record(lambda cs: enqueue_dp_part(cs, lo_, hi_, 2))))
                        if b > a:
                            prog.append(("reduce", (a, b)))
                            prog.append(("opt", record(lambda cs: enqueue_dp_opt(cs, blo, bhi))))
                    prog.append(("join", torch.cuda.Event()))
                    prog.append(("join_opt", torch.cuda.Event()))
                elif mode == "branch":      # ONE graph with the side stream as parallel branches
                    g0 = record(lambda cs: (enqueue_head(cs), enqueue_backward(cs)))
                    check(lib.vs_unet_flip_weight_set(hd))      # undo the capture's flip: the replay below flips
                    prog = [("main", g0)]
                else:
                    # LINEAR graphs (the runtime replays those as one batch of queue packets): per range of units one graph
                    # of the caller's-stream kernels and one of the weight-gradient / optimiser kernels, the second stream
                    # one range behind the first, ordinary events between them
                    cuts = list(range(nu, 0, -seg)) + [0]
                    prog = []
                    for i, (hi_, lo_) in enumerate(zip(cuts[:-1], cuts[1:])):
                        if i == 0:
                            prog.append(("main", record(lambda cs: (enqueue_head(cs), enqueue_part(cs, lo_, hi_, 1)))))
                        else:
                            prog.append(("main", record(lambda cs: enqueue_part(cs, lo_, hi_, 1))))
                        prog.append(("fork", torch.cuda.Event()))
                        prog.append(("side", record(lambda cs: enqueue_part(cs, lo_, hi_, 2))))
                    prog.append(("join", torch.cuda.Event()))
                st["graphs"][wset] = prog
            if self._step_side is None:
                self._step_side = self._library_side_stream(hd, 0)
            cur = torch.cuda.current_stream()
            for op, obj in prog:
                if op == "main":
                    check(lib.vs_graph_launch(obj, stream))
                elif op == "side":
                    check(lib.vs_graph_launch(obj, self._step_side.cuda_stream))
                elif op == "fork":
                    obj.record(cur)
                    self._step_side.wait_event(obj)
                elif op == "reduce":
                    import torch.distributed as dist
                    if self._dp_side is None:
                        self._dp_side = self._library_side_stream(hd, 1)
                    a, b = obj
                    from . import dist as vdist
                    comm = vdist.vs_comm(self.device)
                    if comm is not None:      # the C ABI's communicator: stream-ordered on the optimiser stream, behind the
                        self._dp_side.wait_stream(self._step_side)       # bucket's weight gradients
                        with torch.cuda.stream(self._dp_side):
                            comm.allreduce_sum_(self._flat_grad[a:b])
                            self._flat_grad[a:b].div_(self._world())
                    else:
                        with torch.cuda.stream(self._step_side):      # the collective is ordered behind the bucket's weight gradients
                            h = dist.all_reduce(self._flat_grad[a:b], group=self.dp_group, async_op=True)
                        with torch.cuda.stream(self._dp_side):
                            h.wait()
                            self._flat_grad[a:b].div_(self._world())
                elif op == "opt":
                    check(lib.vs_graph_launch(obj, self._dp_side.cuda_stream))
                elif op == "join_opt":
                    obj.record(self._dp_side)
                    cur.wait_event(obj)
                else:
                    obj.record(self._step_side)
                    cur.wait_event(obj)
            check(lib.vs_unet_flip_weight_set(hd))   # neither a replay nor a split-role recording runs the library's flip
        self._train_forward_token += 1
        self._bnver += 1
        self._wver += 1
        plan["prep"] = (self._flat._version, self._wver, True, None)   # the next forward's weight copies are in place
        opt.step_count += 1
        opt._opt_called = True          # lr_scheduler.step() checks that an optimiser step preceded it
        if not st.get("grads_attached"):
            self._attach_grads(accumulate=False)    # param.grad = views of the flat gradient buffer the step writes
            st["grads_attached"] = True
        return st["loss"].clone() if clone_loss else st["loss"]

    def _library_side_stream(self, handle, index: int):
        """The library's own side stream `index`, checked to sit on another hardware queue than the current stream (HIP multiplexes streams
        onto four queues by default; a torch.cuda.Stream() made here lands on the caller's queue for every second model of a process and
        then runs IN ORDER with it - csrc/unet.hip: acquire_side_streams)."""
        out = _lib.C.c_void_p()
        check(lib.vs_unet_side_stream(handle, _lib.stream_ptr(), index, _lib.C.byref(out)))
        return torch.cuda.ExternalStream(out.value, device=self.device)

    def _world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.dp_group) if self.dp_group is not None else 1

    def _bucket_plan(self, handle):
        """Unit ranges (top of the network first) and the flat-gradient slice each one completes: decoder + head, layer4,
        layer3, then stem + layer1 + layer2.  Backward fills the flat buffer from its end towards its start."""
        names = _lib.unit_names(handle)
        cuts = [len(names)]
        prefixes = ("decoder.", "encoder.layer4.0.", "encoder.layer3.0.")
        if self.encoder_name.startswith("efficientnet"):     # the MBConv stages behind the 1/32 and 1/16 features (smp's stage_idxs)
            s16, s32 = {"efficientnet-b3": (8, 18), "efficientnet-b4": (10, 22)}[self.encoder_name]
            prefixes = ("decoder.", f"encoder._blocks.{s32}.", f"encoder._blocks.{s16}.")
        for prefix in prefixes:
            cuts.append(next(i for i, nm in enumerate(names) if nm.startswith(prefix)))
        cuts.append(0)
        plan = []
        for hi, lo in zip(cuts[:-1], cuts[1:]):
            a = lib.vs_unet_unit_param_offset(handle, lo)
            b = lib.vs_unet_unit_param_offset(handle, hi)
            plan.append((lo, hi, a, b))
        return plan

    def _backward_bucketed(self, plan, x, dlogits, n, need_enc, fused=None):
        """Data-parallel backward: every bucket's all-reduce (RCCL, async) overlaps with the backward of the layers below it.
        With ``fused`` (a FusedAdamW) each bucket's optimiser step and next-forward weight copies follow its all-reduce on a
        side stream, so neither sits on the critical path (same numbers as all-reduce, then step(), then prepare)."""
        import torch.distributed as dist
        if "buckets" not in plan:
            plan["buckets"] = self._bucket_plan(plan["handle"])
        from . import dist as vdist
        world = self._world()
        handles = []
        main = torch.cuda.current_stream()
        if self._dp_side is None:
            self._dp_side = self._library_side_stream(plan["handle"], 1)
        side = self._dp_side
        comm = vdist.vs_comm(self.device)     # VOLSEG_COMM=rccl: the C ABI's communicator, stream-ordered on the side stream
        low = None
        if self.dp_grad_dtype != torch.float32:
            comm = None                        # (the C ABI's all-reduce is fp32)
            if plan.get("grad_low") is None or plan["grad_low"].dtype != self.dp_grad_dtype:
                plan["grad_low"] = torch.empty(self._flat_grad.numel(), dtype=self.dp_grad_dtype, device=self._flat_grad.device)
            low = plan["grad_low"]
        for lo, hi, a, b in plan["buckets"]:
            check(lib.vs_unet_backward_range(plan["handle"], ptr(self._flat), ptr(x), ptr(dlogits), n, 1 if need_enc else 0,
                                             ptr(self._flat_grad), ptr(plan["ws"]), _lib.stream_ptr(), lo, hi))
            if b > a:
                if comm is not None:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        comm.allreduce_sum_(self._flat_grad[a:b])
                    handles.append((None, lo, hi, a, b))
                elif low is not None:
                    # reduced-precision transport (dp_grad_dtype = bfloat16): the bucket travels as bf16 - half the bytes
                    # through the per-link-bound xGMI ring - and comes back into the fp32 buffer; master gradients,
                    # AdamW and the weights stay fp32
                    low[a:b].copy_(self._flat_grad[a:b])
                    handles.append((dist.all_reduce(low[a:b], group=self.dp_group, async_op=True), lo, hi, a, b))
                else:
                    handles.append((dist.all_reduce(self._flat_grad[a:b], group=self.dp_group, async_op=True), lo, hi, a, b))
        if fused is None:
            for h, lo, hi, a, b in handles:
                if h is not None:
                    h.wait()
                if low is not None:
                    self._flat_grad[a:b].copy_(low[a:b])
            main.wait_stream(side)
            self._flat_grad.div_(world)
            return
        g = fused.param_groups[0]
        mask = fused._grad_mask_for(self, need_enc)
        with torch.cuda.stream(side):
            for h, lo, hi, a, b in handles:
                if h is not None:
                    h.wait()                               # this (side) stream waits for the bucket's all-reduce
                if low is not None:
                    self._flat_grad[a:b].copy_(low[a:b])
                self._flat_grad[a:b].div_(world)
                check(lib.vs_adamw_step(ptr(self._flat) + 4 * a, ptr(self._flat_grad) + 4 * a, ptr(fused.exp_avg) + 4 * a,
                                        ptr(fused.exp_avg_sq) + 4 * a, (ptr(mask) + a) if mask is not None else None, b - a,
                                        float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                        float(g["weight_decay"]), fused.step_count + 1, _lib.stream_ptr()))
                check(lib.vs_unet_prepare_range(plan["handle"], ptr(self._flat), ptr(plan["ws"]), _lib.stream_ptr(), lo, hi))
        main.wait_stream(side)
        check(lib.vs_unet_flip_weight_set(plan["handle"]))

    def _allreduce_grads(self):
        import torch.distributed as dist

        world = dist.get_world_size(self.dp_group)
        if world == 1:
            return
        from . import dist as vdist
        comm = vdist.vs_comm(self.device)
        if self.dp_grad_dtype == torch.float32 and comm is not None:
            comm.allreduce_sum_(self._flat_grad)
            self._flat_grad.div_(world)
        elif self.dp_grad_dtype == torch.float32:
            dist.all_reduce(self._flat_grad, group=self.dp_group)
            self._flat_grad.div_(world)
        else:
            g = self._flat_grad.to(self.dp_grad_dtype)
            dist.all_reduce(g, group=self.dp_group)
            self._flat_grad.copy_(g)
            self._flat_grad.div_(world)

    def forward(self, x):
        x = self._check_input(x)
        if self.training and self.precision == "fp16":
            # fp16 activations and weights (BASELINE configs[4]) are the INFERENCE precision: batch statistics, the backward pass
            # and the optimiser are built for fp32 / bf16 (no loss scaling here) - train in bf16 and predict in fp16
            raise RuntimeError("precision 'fp16' is inference only: call model.eval(), or train with precision 'bf16' / 'fp32'")
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if self._anchor is None or self._anchor.device != self.device:
                self._anchor = torch.zeros((), device=self.device, requires_grad=True)
            return _UnetFn.apply(x, self, self._anchor)
        return self._forward_impl(x, training=self.training)

    # ------------------------------------------------------------------ optimizer
    def fused_adamw(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                    fuse_step_into_backward: bool = False):
        return FusedAdamW(self, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                          fuse_step_into_backward=fuse_step_into_backward)


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (the reference's optimiser, vol_seg_2d_trainer.py:395-396) as ONE HIP
    kernel over the model's flat parameter buffer.  Exposes a normal ``param_groups[0]`` with ``lr`` and
    ``betas`` so LambdaLR / OneCycleLR (which cycles beta1, :401-408) drive it unchanged."""

    def __init__(self, model: VolSegUnet, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 fuse_step_into_backward: bool = False):
        """``fuse_step_into_backward``: the update of every tensor is queued inside ``loss.backward()`` right behind that
        tensor's gradient (vs_unet_backward_adamw), hidden under the rest of the backward pass; ``step()`` then only does
        the bookkeeping.  Same numbers as the unfused path.  It assumes the reference's loop - exactly one ``step()`` per
        ``backward()`` with the learning rate set before ``backward()`` (vol_seg_2d_trainer.py:419-432) - and falls back
        to the plain path for data-parallel runs or unusual ``requires_grad`` patterns."""
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(list(model.parameters()), defaults)
        self.model = model
        self._stepped_in_backward = False
        model._fused_optimizer = self if fuse_step_into_backward else None
        # recorded steps (fused_train_step) bake in the moment buffers and hyper-parameter block of the optimiser they were
        # recorded with: a new optimiser on the same model must never replay them
        model._drop_steps()
        self.exp_avg = torch.zeros_like(model._flat)
        self.exp_avg_sq = torch.zeros_like(model._flat)
        self.step_count = 0
        self._mask = None
        self._mask_key = None

    def _can_fuse(self, model, need_enc: bool) -> bool:
        if self._stepped_in_backward:
            raise RuntimeError("FusedAdamW(fuse_step_into_backward=True): backward() called twice without step() - gradient "
                               "accumulation needs fuse_step_into_backward=False")
        for p, _, _, _, enc in model._param_cache:
            if p.requires_grad != (need_enc or not enc):
                return False   # only "everything" or "everything but the encoder convolutions" can be fused
        return True

    def _grad_mask_for(self, model, need_enc: bool):
        """uint8 mask over the flat buffer for the data-parallel fused step: None (everything trains) or zeros on the frozen
        encoder convolutions (_can_fuse has already established that these are the only two patterns)."""
        unused = [t for t in model._table if t[2] <= KIND_BIAS and t[0].startswith(model.UNUSED_PREFIXES)]
        if need_enc and not unused:
            return None
        key = "_frozen_mask" if not need_enc else "_unused_mask"
        if getattr(self, key, None) is None:
            m = torch.ones(model._flat.numel(), dtype=torch.uint8)
            for (name, shape, kind, off) in [t for t in model._table if t[2] <= KIND_BIAS]:
                if (not need_enc and "encoder" in name and "conv" in name) or name.startswith(model.UNUSED_PREFIXES):
                    m[off:off + math.prod(shape)] = 0
            setattr(self, key, m.to(model.device))
        return getattr(self, key)

    def _grad_mask(self):
        params = list(self.model.parameters())
        key = tuple(p.requires_grad and p.grad is not None for p in params)
        if key != self._mask_key:
            if all(key):
                self._mask = None
            else:
                m = torch.zeros(self.model._flat.numel(), dtype=torch.uint8)
                for (name, shape, kind, off), on in zip([t for t in self.model._table if t[2] <= KIND_BIAS], key):
                    if on:
                        m[off:off + math.prod(shape)] = 1
                self._mask = m.to(self.model.device)
            self._mask_key = key
        return self._mask

    def zero_grad(self, set_to_none: bool = True):
        # gradients live in the model's flat buffer and are overwritten by every backward: keep the
        # aliases (no per-tensor work); set_to_none=False zeroes the flat buffer in one kernel
        if not set_to_none and self.model._flat_grad is not None:
            self.model._flat_grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        m = self.model
        if m._flat_grad is None:
            return loss
        g = self.param_groups[0]
        self.step_count += 1
        if self._stepped_in_backward:   # the kernels already ran inside backward()
            self._stepped_in_backward = False
            return loss
        mask = self._grad_mask()
        check(lib.vs_adamw_step(ptr(m._flat), ptr(m._flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(mask),
                                m._flat.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                float(g["eps"]), float(g["weight_decay"]), self.step_count, _lib.stream_ptr()))
        m._wver += 1  # weight copies are re-derived before the next forward
        return loss

    def _param_table(self):
        return [t for t in self.model._table if t[2] <= KIND_BIAS]   # order of model.parameters()

    def state_dict(self):
        """torch.optim.AdamW's own format - per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` with the parameter's
        (OIHW) shape, indexed by position - so ``optimizer_state_dict`` of a checkpoint interchanges with the reference's
        ``_load_in_weights(optimizer=True)`` (vol_seg_2d_trainer.py:276-296)."""
        m = self.model
        state = {}
        params = list(m.parameters())
        for i, (name, shape, kind, off) in enumerate(self._param_table()):
            # torch.optim.AdamW holds no state for a parameter that never received a gradient (the frozen encoder
            # convolutions of the reference's first phase, tensors the forward never reads): no entry, so that a later
            # unfreeze starts its bias correction from step 1 of THAT parameter on the reference side
            if self.step_count == 0 or (i < len(params) and not params[i].requires_grad) or name.startswith(m.UNUSED_PREFIXES):
                continue
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": m._view_of(self.exp_avg, shape, kind, off).clone(memory_format=torch.contiguous_format),
                        "exp_avg_sq": m._view_of(self.exp_avg_sq, shape, kind, off).clone(memory_format=torch.contiguous_format)}
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d.setdefault("amsgrad", False); d.setdefault("maximize", False); d.setdefault("foreach", None)
            d.setdefault("capturable", False); d.setdefault("differentiable", False); d.setdefault("fused", None)
            d["params"] = list(range(len(self._param_table())))
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        st = sd["state"]
        if "exp_avg" in st:      # round-1 flat format
            self.step_count = int(st["step"])
            self.exp_avg.copy_(st["exp_avg"])
            self.exp_avg_sq.copy_(st["exp_avg_sq"])
        else:                    # torch.optim.AdamW format (ours, or a reference checkpoint); absent entries = never stepped
            m = self.model
            steps = [int(float(v["step"])) for v in st.values() if "step" in v]
            self.step_count = max(steps) if steps else 0
            self.exp_avg.zero_(); self.exp_avg_sq.zero_()
            with torch.no_grad():
                for i, (name, shape, kind, off) in enumerate(self._param_table()):
                    e = st.get(i, st.get(str(i)))
                    if e is None:
                        continue
                    m._view_of(self.exp_avg, shape, kind, off).copy_(e["exp_avg"])
                    m._view_of(self.exp_avg_sq, shape, kind, off).copy_(e["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s.items() if k != "params"})


def load_oracle_state(model: VolSegUnet, state_dict) -> None:
    """Strict load of an smp-keyed state dict (e.g. from the CPU oracle or a reference checkpoint)."""
    missing, unexpected = model.load_state_dict(state_dict, strict=True)
    assert not missing and not unexpected
    logging.debug("loaded %d tensors", len(state_dict))
