"""FlatPose2Mesh -- the module the reference's scripts instantiate (lib/models/pose2mesh_net.py).

    model = pose2mesh_net.get_model(num_joint, graph_L)          # base.py:57, demo/run.py:120
    cam_mesh, pose3d = model(pose2d)                             # (B,V0,3) metres, (B,J,3) millimetres

State-dict keys are `pose_lifter.*` / `pose2mesh.*` exactly as in the reference, so its
final.pth.tar checkpoints load with load_state_dict().
"""
import torch
import torch.nn as nn

from . import meshnet, posenet


class FlatPose2Mesh(nn.Module):
    def __init__(self, num_joint, graph_L, mano=None, posenet_pretrained=None, posenet_path=None):
        """posenet_pretrained / posenet_path: default = cfg.MODEL.posenet_pretrained / cfg.MODEL.posenet_path of the
        reference's config when its scripts have loaded it (lib/models/pose2mesh_net.py:13, lib/models/posenet.py:89-92
        -- every training yaml sets posenet_pretrained: True), else False."""
        super().__init__()
        self.num_joint = num_joint
        if posenet_pretrained is None:
            posenet_pretrained = posenet.cfg_posenet_pretrained()
        self.pose_lifter = posenet.get_model(num_joint, hid_dim=4096, num_layer=2, p_dropout=0.5,
                                             pretrained=posenet_pretrained, posenet_path=posenet_path)
        self.pose2mesh = meshnet.get_model(num_joint_input_chan=2 + 3, num_mesh_output_chan=3, graph_L=graph_L,
                                           mano=mano)

    def set_inference(self, real_only=True, perm_reverse=None, nv=None, scale=1.0):
        """See Pose2Mesh.set_inference (eval() + no_grad fast path: real vertices only, optional mesh-order output)."""
        self.pose2mesh.set_inference(real_only, perm_reverse, nv, scale)
        return self

    def accumulate_grads_in_place(self, enable=True):
        """Both halves add their gradients straight into the parameters' .grad tensors (Pose2Mesh / LinearModel
        .accumulate_grads_in_place: for training loops that own a flat gradient buffer, optim.FlatAdam / FlatRMSprop)."""
        self.pose2mesh.accumulate_grads_in_place(enable)
        self.pose_lifter.accumulate_grads_in_place(enable)
        return self

    def set_grad_sink(self, sink):
        """dist.BucketedAllReduce.notify for the gradients that bypass autograd (see Pose2Mesh.set_grad_sink)."""
        self.pose2mesh.set_grad_sink(sink)
        self.pose_lifter.set_grad_sink(sink)
        return self

    def forward(self, pose2d):
        pose3d = self.pose_lifter(pose2d.view(len(pose2d), -1)).reshape(-1, self.num_joint, 3)
        # MeshNet gets no gradient path into PoseNet (pose2mesh_net.py:19)
        pose_combine = torch.cat((pose2d, pose3d.detach() / 1000), dim=2)
        return self.pose2mesh(pose_combine), pose3d


def get_model(num_joint, graph_L, mano=None, posenet_pretrained=None, posenet_path=None):
    """lib/models/pose2mesh_net.py:25-28."""
    return FlatPose2Mesh(num_joint, graph_L, mano=mano, posenet_pretrained=posenet_pretrained, posenet_path=posenet_path)
