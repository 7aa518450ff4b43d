"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the demo's 2D-pose preprocessing (BASELINE configs[0]:
demo/run.py single pose h36m_joint_input.npy -> mesh).  Only tests/ may import this file.

Restates:
  get_bbox / process_bbox        lib/coord_utils.py:21-39, 42-66
  get_center_scale               lib/coord_utils.py:7-18
  get_affine_transform           lib/aug_utils.py:140-175  (cv2.getAffineTransform = exact 3-point solve, float64)
  affine_transform               lib/aug_utils.py:178-181
  j2d_processing                 lib/aug_utils.py:51-64    (no flip)
  model-input normalisation      demo/run.py:149-160 (= data/Human36M/dataset.py:383-388)
  mesh epilogue                  demo/run.py:169-171

Parity pinning: tests/golden/demo_h36m.npz was produced by the REAL functions (imported through
oracle/ref_loader.load_aug(); OpenCV is not installed, so the faked cv2 module carries the same 3-point solve);
tests/test_oracle.py checks this restatement against it, including the reference's integer truncation quirk
(kp[i, :2] = affine_transform(...) writes into the int64 fixture array, lib/aug_utils.py:59).
"""
import numpy as np

INPUT_SHAPE = (384, 288)     # cfg.MODEL.input_shape (lib/core/config.py:52): (height, width)


def get_bbox(joint_img):
    """coord_utils.py:21-39."""
    x_img, y_img = joint_img[:, 0], joint_img[:, 1]
    xmin, ymin, xmax, ymax = min(x_img), min(y_img), max(x_img), max(y_img)
    x_center = (xmin + xmax) / 2.
    width = xmax - xmin
    xmin, xmax = x_center - 0.5 * width, x_center + 0.5 * width
    y_center = (ymin + ymax) / 2.
    height = ymax - ymin
    ymin, ymax = y_center - 0.5 * height, y_center + 0.5 * height
    return np.array([xmin, ymin, xmax - xmin, ymax - ymin]).astype(np.float32)


def process_bbox(bbox, aspect_ratio=None, scale=1.0):
    """coord_utils.py:42-66."""
    x, y, w, h = bbox
    x1, y1, x2, y2 = x, y, x + (w - 1), y + (h - 1)
    if w * h > 0 and x2 >= x1 and y2 >= y1:
        bbox = np.array([x1, y1, x2 - x1, y2 - y1])
    else:
        return None
    w, h = bbox[2], bbox[3]
    c_x, c_y = bbox[0] + w / 2., bbox[1] + h / 2.
    if aspect_ratio is None:
        aspect_ratio = INPUT_SHAPE[1] / INPUT_SHAPE[0]
    if w > aspect_ratio * h:
        h = w / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    bbox[2] = w * scale
    bbox[3] = h * scale
    bbox[0] = c_x - bbox[2] / 2.
    bbox[1] = c_y - bbox[3] / 2.
    return bbox


def get_center_scale(box_info):
    """coord_utils.py:7-18."""
    x, y, w, h = box_info
    center = np.zeros((2), dtype=np.float32)
    center[0] = x + w * 0.5
    center[1] = y + h * 0.5
    scale = np.array([w * 1.0, h * 1.0], dtype=np.float32)
    return center, scale


def _get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]     # aug_utils.py:188-195


def _get_3rd_point(a, b):
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)                              # aug_utils.py:182-184


def three_point_affine(src, dst):
    """What cv2.getAffineTransform computes: M (2x3, float64) with M [x y 1]^T = dst for the 3 point pairs."""
    A = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(A, np.asarray(dst, np.float64)).T


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32)):
    """aug_utils.py:140-175 (inv = 0)."""
    src_w = scale[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = _get_dir([0, src_w * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale * shift
    src[1, :] = center + src_dir + scale * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2:, :] = _get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = _get_3rd_point(dst[0, :], dst[1, :])
    return three_point_affine(np.float32(src), np.float32(dst))


def affine_transform(pt, t):
    new_pt = np.array([pt[0], pt[1], 1.]).T                                                     # aug_utils.py:178-181
    return np.dot(t, new_pt)[:2]


def j2d_processing(kp, res, bbox, rot):
    """aug_utils.py:51-64 without the flip.  NOTE: `kp` keeps its dtype while the transformed points are written
    back (:59), so an integer fixture (demo/h36m_joint_input.npy is int64) is truncated towards zero before the
    final astype('float32') (:63) -- the reference's behaviour, preserved."""
    center, scale = get_center_scale(bbox)
    trans = get_affine_transform(center, scale, rot, res)
    for i in range(kp.shape[0]):
        kp[i, :2] = affine_transform(kp[i, :2].copy(), trans)
    return kp.astype('float32'), trans


def demo_model_input(joint_input):
    """demo/run.py:149-160: (J, 2+) pixel joints -> (J, 2) float32 standardised model input."""
    bbox = get_bbox(joint_input)
    bbox2 = process_bbox(bbox.copy())
    joint_img, _ = j2d_processing(joint_input.copy(), (INPUT_SHAPE[1], INPUT_SHAPE[0]), bbox2, 0)
    joint_img = joint_img[:, :2]
    joint_img /= np.array([[INPUT_SHAPE[1], INPUT_SHAPE[0]]])
    mean, std = np.mean(joint_img, axis=0), np.std(joint_img, axis=0)
    joint_img = (joint_img.copy() - mean) / std
    return joint_img.astype(np.float32), bbox, bbox2        # torch.Tensor(...) at run.py:160 makes it float32


def demo_mesh_epilogue(pred_mesh, graph_perm_reverse, nv, joint_regressor):
    """demo/run.py:169-171: tree order -> mesh-model vertex order, then joints = J_regressor @ mesh."""
    mesh = np.asarray(pred_mesh)[:, np.asarray(graph_perm_reverse)[:nv], :]
    return mesh, np.matmul(np.asarray(joint_regressor)[None], mesh)
