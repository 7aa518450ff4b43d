"""Graph-captured inference (BASELINE configs[1] / the Tester's loop, lib/core/base.py:196-204).

A batch-64 forward is ~190 kernel launches of 5-50 us each: launch-bound when issued one by one from Python.  The whole
step -- FlatPose2Mesh in eval(), the real-vertices-only fast path (Pose2Mesh.set_inference) with the mesh written straight
in mesh-model vertex order, and the sparse joint regression -- is captured once into a hipGraph (torch.cuda.CUDAGraph on
ROCm) and replayed per batch.  Shapes are static per instance; the input is copied into a resident buffer.
"""
import ctypes as _ct

import numpy as _np
import torch

from . import _lib
from . import loss as _loss
from . import ops as _ops


class GraphedInference:
    """step = GraphedInference(model, perm_reverse, nv, joint_regressor, batch, scale=1000.0)
    mesh, joints, pose3d = step(pose2d)        # [B, nv, 3] (mesh order, x scale), [B, J, 3], [B, J, 3]
    The returned tensors are the graph's static outputs: they are overwritten by the next call.

    The model is put in eval() and switched to the mesh-order inference output (Pose2Mesh.set_inference) for good: other
    users of the same module object see that layout too.  The captured graph holds raw pointers to the derived weight
    operands (packed / split weights, eval BatchNorm coefficients) that existed at capture time; the instance keeps
    those tensors alive, and when the weights change afterwards (optimizer steps, load_state_dict, a train-mode forward:
    ops.WEIGHT_EPOCH, or the (pointer, version) of ANY parameter or buffer moves - checked on every call) the next call
    re-captures instead of replaying stale operands.  (Edits made through `tensor.data` do not move Tensor._version - torch
    gives `.data` its own counter -: a caller that writes weights that way calls ops.bump_weight_epoch() afterwards.)"""

    def __init__(self, model, perm_reverse, nv, joint_regressor, batch, scale=1000.0, use_graph=True, warmup=3):
        p = next(model.parameters())
        if not p.is_cuda:
            raise _lib.P2MError("GraphedInference needs the model on the GPU")
        self.device = p.device
        self.model = model.eval()
        model.set_inference(real_only=True, perm_reverse=perm_reverse, nv=nv, scale=scale)
        self.nv, self.J = int(nv), int(_np.asarray(joint_regressor).shape[0])
        self._reg = {k: torch.from_numpy(_np.ascontiguousarray(v)).to(self.device)
                     for k, v in _loss._regressor_tables(_np.asarray(joint_regressor, dtype=_np.float32), nv).items()}
        self._ident = torch.arange(self.nv, dtype=torch.int32, device=self.device)
        num_joint = model.num_joint
        self.pose2d = torch.zeros((batch, num_joint, 2), device=self.device, dtype=torch.float32)
        self.graph, self.use_graph, self.warmup = None, bool(use_graph), max(1, int(warmup))
        self.captures = 0
        self._capture()

    # Checked on EVERY call, exactly: ops.WEIGHT_EPOCH (bumped by the flat optimizers, every train-mode forward and
    # GraphedTrainStep, which move weights by raw pointer) and the (pointer, version) of EVERY parameter and buffer - an
    # in-place edit of one interior tensor (`with torch.no_grad(): net.cl[7].weight.copy_(...)`, a partial load_state_dict)
    # moves nothing else.  NOT caught: writes through `tensor.data` (torch gives `.data` its own version counter) - after
    # such a write call ops.bump_weight_epoch(), or the graph replays the stale packed weights (class docstring).  The tensor list is cached at capture time, so the check is one pass of two attribute reads per tensor (~40 us
    # for the ~190 tensors of FlatPose2Mesh; the 0.2 ms a naive check costs is the model.parameters() walk, not this).
    def _weights_tag(self):
        return (_ops.WEIGHT_EPOCH,) + tuple((t.data_ptr(), t._version) for t in self._tensors)

    def _weights_moved(self):
        if _ops.WEIGHT_EPOCH != self._tag[0]:
            return True
        tag = self._tag
        i = 1
        for t in self._tensors:
            if (t.data_ptr(), t._version) != tag[i]:
                return True
            i += 1
        return False

    def _capture(self):
        if self.model.training:
            # a capture in train() would bake batch statistics into the graph and let every replay move the running
            # statistics; the instance owns the module's mode (see the class docstring), so put it back
            self.model.eval()
        with torch.cuda.device(self.device), torch.no_grad():
            self.graph = None
            params, bufs = list(self.model.parameters()), list(self.model.buffers())
            self._tensors = params + bufs
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):              # warm-up: graph handles, weight packs, allocator pools
                for _ in range(self.warmup):
                    out = self._eager()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            # the operands the captured kernels will read: referenced here so that a later rebuild of the cache entries
            # cannot hand their memory to someone else while the graph still points at it
            self._held = [m._weight_cache._d.copy() for m in self.model.modules() if hasattr(m, "_weight_cache")]
            if self.use_graph:
                g = torch.cuda.CUDAGraph()
                with _ops.capture_guard(), torch.cuda.graph(g):     # (no finalizer may run inside the capture)
                    out = self._eager()
                self.graph = g
            self.mesh, self.joints, self.pose3d = out
            self._tag = self._weights_tag()
            self.captures += 1

    def _eager(self):
        mesh, pose3d = self.model(self.pose2d)                       # mesh: [B, nv, 3] mesh order x scale
        B = mesh.shape[0]
        joints = torch.empty((B, self.J, 3), device=self.device, dtype=torch.float32)

        def p(t):
            return _ct.c_void_p(t.data_ptr())
        # joints = J_regressor @ mesh (base.py:204): CSR mat-vec over the already-ordered mesh (perm = identity)
        _lib.check(_lib.hip().p2m_mesh_epilogue(p(mesh), self.nv, p(self._ident), self.nv, 1.0, p(self._reg["jr_ptr"]),
                                                p(self._reg["jr_idx"]), p(self._reg["jr_val"]), self.J, None, p(joints),
                                                B, _ct.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "p2m_mesh_epilogue")
        return mesh, joints, pose3d

    @torch.no_grad()
    def __call__(self, pose2d):
        self.pose2d.copy_(pose2d.reshape(self.pose2d.shape), non_blocking=True)
        if self.graph is not None and (self.model.training or self._weights_moved()):
            self._capture()                            # the weights moved since the capture: its operands are stale
                                                       # (or someone left the module in train(): re-capture in eval())
        if self.graph is not None:
            self.graph.replay()
            return self.mesh, self.joints, self.pose3d
        with torch.cuda.device(self.device):
            self.mesh, self.joints, self.pose3d = self._eager()
        return self.mesh, self.joints, self.pose3d
