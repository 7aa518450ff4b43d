"""pose2mesh_release_amd -- MI355X (gfx950) native implementation of the Pose2Mesh coarse-to-fine
Chebyshev graph-convolution hot path, behind the reference's own module-factory API.

    from pose2mesh_release_amd import graph_utils, meshnet, pose2mesh_net
    _, graph_L, _, perm_reverse = graph_utils.build_coarse_graphs(faces, J, skeleton, flip_pairs, levels=9)
    model = pose2mesh_net.get_model(J, graph_L).cuda()
    cam_mesh, pose3d = model(pose2d)          # same I/O as lib/models/pose2mesh_net.py:16-22
"""
__version__ = "0.1.0"
