// Mesh losses of the Pose2Mesh train step, forward AND gradient, in four small launches (gfx950).
//
// Reference arithmetic (stock PyTorch there, ~100 tiny kernels + sort-based index_put backward, ~10 ms at B=256):
//   pred_mesh = cam_mesh[:, graph_perm_reverse[:nv], :]                      lib/core/base.py:130
//   pred_pose = J_regressor @ (pred_mesh * 1000)                             lib/core/base.py:131
//   CoordLoss (masked mean L1)        on pred_mesh / gt_mesh, pred_pose / gt_pose   lib/core/loss.py:10-23
//   NormalVectorLoss                                                         lib/core/loss.py:62-88
//   EdgeLengthLoss                                                           lib/core/loss.py:91-114
//   loss = L1_mesh + w_n * normal + w_e * edge + w_j * L1_pose               lib/core/base.py:134-143
// Everything here is a gather over 13 776 faces / 6 890 vertices per sample: HBM/L2-bound, a few MB.
// The gradient w.r.t. cam_mesh (tree order, zeros on fake vertices) is produced directly; summation orders
// are fixed (per-face gradients are gathered per vertex through a vertex->corner CSR, no atomics).
#include "p2m_common.h"

namespace p2m {

struct LossArgs {
  const float* cam;      // [B, V0, 3]
  const int* perm;       // [nv]
  const float* gt_mesh;  // [B, nv, 3]
  const float* valid_mesh;   // [B, nv] or NULL
  const int* faces;      // [F, 3]
  const int* vf_ptr;     // [nv + 1]
  const int* vf_idx;     // [3F]   face*3 + corner
  const int* jr_ptr;     // [J + 1]   CSR of the joint regressor (107 non-zeros for the 17 x 6890 h36m regressor)
  const int* jr_idx;     // [nnz]     vertex (mesh-model order)
  const float* jr_val;   // [nnz]
  const int* vj_ptr;     // [nv + 1]  the same matrix by vertex (CSC): joints that use vertex v
  const int* vj_idx;     // [nnz]     joint
  const float* vj_val;   // [nnz]
  const float* gt_pose;  // [B, J, 3]
  const float* valid_pose;   // [B, J] or NULL
  float* face_grad;      // [B, F, 9]
  float* pose_sign;      // [B, J, 3]
  float* partial;        // [4][npart]
  float* grad_cam;       // [B, V0, 3]
  int B, V0, nv, F, J, npart;
  float s_vertex, s_normal, s_edge, s_joint;   // weight / element count
};

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 scale(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 normalize(V3 a, float* norm_out = nullptr) {   // F.normalize(p=2, eps=1e-12)
  float n = sqrtf(dot(a, a));
  if (norm_out) *norm_out = n;
  return scale(a, 1.f / fmaxf(n, 1e-12f));
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += sh[i];
  __syncthreads();
  return s;   // valid in thread 0
}

// one thread per (b, j): regressed joint over the CSR row (6 entries per joint), L1 term and the sign needed by the
// vertex gradient
__global__ __launch_bounds__(256) void k_pose_regress(LossArgs a) {
  __shared__ float sh[4];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  float loss = 0.f;
  if (idx < a.B * a.J) {
    const int b = idx / a.J, j = idx - b * a.J;
    const float* cam = a.cam + (long)b * a.V0 * 3;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int k = a.jr_ptr[j]; k < a.jr_ptr[j + 1]; k++) {
      const float w = a.jr_val[k];
      const float* p = cam + (long)a.perm[a.jr_idx[k]] * 3;
      s0 = fmaf(w, p[0] * 1000.f, s0);
      s1 = fmaf(w, p[1] * 1000.f, s1);
      s2 = fmaf(w, p[2] * 1000.f, s2);
    }
    const float val = a.valid_pose ? a.valid_pose[idx] : 1.f;
    const float pose[3] = {s0, s1, s2};
    for (int c = 0; c < 3; c++) {
      const float d = val * pose[c] - val * a.gt_pose[(long)idx * 3 + c];
      loss += fabsf(d);
      a.pose_sign[(long)idx * 3 + c] = a.s_joint * val * sgn(d) * 1000.f;
    }
  }
  const float sl = block_sum(loss, sh);
  if (threadIdx.x == 0) a.partial[3 * a.npart + blockIdx.x] = sl * a.s_joint;
}

// one thread per (b, face): normal + edge terms and their gradients w.r.t. the three corners
__global__ __launch_bounds__(256) void k_face_terms(LossArgs a) {
  __shared__ float sh[4];
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  float l_normal = 0.f, l_edge = 0.f;
  if (idx < (long)a.B * a.F) {
    const int b = (int)(idx / a.F), f = (int)(idx % a.F);
    const int i0 = a.faces[f * 3], i1 = a.faces[f * 3 + 1], i2 = a.faces[f * 3 + 2];
    const float* cam = a.cam + (long)b * a.V0 * 3;
    const float* gt = a.gt_mesh + (long)b * a.nv * 3;
    const V3 p0 = ld3(cam + (long)a.perm[i0] * 3), p1 = ld3(cam + (long)a.perm[i1] * 3), p2 = ld3(cam + (long)a.perm[i2] * 3);
    const V3 g0 = ld3(gt + (long)i0 * 3), g1 = ld3(gt + (long)i1 * 3), g2 = ld3(gt + (long)i2 * 3);
    // ---- normal-vector loss (loss.py:70-87)
    float n1, n2, n3;
    const V3 e1 = normalize(sub(p1, p0), &n1), e2 = normalize(sub(p2, p0), &n2), e3 = normalize(sub(p2, p1), &n3);
    const V3 ng = normalize(cross(normalize(sub(g1, g0)), normalize(sub(g2, g0))));
    const float c1 = dot(e1, ng), c2 = dot(e2, ng), c3 = dot(e3, ng);
    l_normal = fabsf(c1) + fabsf(c2) + fabsf(c3);
    // d|c_k| / d d_k = sign(c_k) * (n - e_k c_k) / |d_k|
    const V3 q1 = scale(sub(ng, scale(e1, c1)), a.s_normal * sgn(c1) / fmaxf(n1, 1e-12f));
    const V3 q2 = scale(sub(ng, scale(e2, c2)), a.s_normal * sgn(c2) / fmaxf(n2, 1e-12f));
    const V3 q3 = scale(sub(ng, scale(e3, c3)), a.s_normal * sgn(c3) / fmaxf(n3, 1e-12f));
    V3 d0 = scale(add(q1, q2), -1.f);       // p0 enters d1, d2 negatively
    V3 d1 = sub(q1, q3);                    // p1: +d1, -d3
    V3 d2 = add(q2, q3);                    // p2: +d2, +d3
    if (a.s_normal == 0.f) d0 = d1 = d2 = V3{0.f, 0.f, 0.f};
    // ---- edge-length loss (loss.py:99-113)
    const V3 u01 = sub(p0, p1), u02 = sub(p0, p2), u12 = sub(p1, p2);
    const float o1 = sqrtf(dot(u01, u01)), o2 = sqrtf(dot(u02, u02)), o3 = sqrtf(dot(u12, u12));
    const V3 h01 = sub(g0, g1), h02 = sub(g0, g2), h12 = sub(g1, g2);
    const float t1 = sqrtf(dot(h01, h01)), t2 = sqrtf(dot(h02, h02)), t3 = sqrtf(dot(h12, h12));
    l_edge = fabsf(o1 - t1) + fabsf(o2 - t2) + fabsf(o3 - t3);
    if (a.s_edge != 0.f) {
      // (w_edge == 0, i.e. before cfg.TRAIN.edge_loss_start, the reference does not evaluate this loss at all,
      // base.py:141-143: no 0/0 from a degenerate predicted edge may leak into the gradient then)
      const V3 r1 = scale(u01, a.s_edge * sgn(o1 - t1) / o1);
      const V3 r2 = scale(u02, a.s_edge * sgn(o2 - t2) / o2);
      const V3 r3 = scale(u12, a.s_edge * sgn(o3 - t3) / o3);
      d0 = add(d0, add(r1, r2));
      d1 = add(d1, sub(r3, r1));
      d2 = sub(d2, add(r2, r3));
    }
    float* fg = a.face_grad + idx * 9;
    fg[0] = d0.x; fg[1] = d0.y; fg[2] = d0.z;
    fg[3] = d1.x; fg[4] = d1.y; fg[5] = d1.z;
    fg[6] = d2.x; fg[7] = d2.y; fg[8] = d2.z;
  }
  const float sn = block_sum(l_normal, sh);
  const float se = block_sum(l_edge, sh);
  if (threadIdx.x == 0) {
    a.partial[1 * a.npart + blockIdx.x] = sn * a.s_normal;
    a.partial[2 * a.npart + blockIdx.x] = se * a.s_edge;
  }
}

// one thread per (b, real vertex): vertex L1 + incident face gradients + joint-regressor term -> grad_cam
__global__ __launch_bounds__(256) void k_vertex_grad(LossArgs a) {
  __shared__ float sh[4];
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  float l_v = 0.f;
  if (idx < (long)a.B * a.nv) {
    const int b = (int)(idx / a.nv), v = (int)(idx % a.nv);
    const float val = a.valid_mesh ? a.valid_mesh[idx] : 1.f;
    const float* p = a.cam + ((long)b * a.V0 + a.perm[v]) * 3;
    const float* g = a.gt_mesh + idx * 3;
    float gr[3];
    for (int c = 0; c < 3; c++) {
      const float d = val * p[c] - val * g[c];
      l_v += fabsf(d);
      gr[c] = a.s_vertex * val * sgn(d);
    }
    const float* fgb = a.face_grad + (long)b * a.F * 9;
    for (int k = a.vf_ptr[v]; k < a.vf_ptr[v + 1]; k++) {
      const float* fg = fgb + (long)a.vf_idx[k] * 3;     // (face*3 + corner) * 3
      gr[0] += fg[0]; gr[1] += fg[1]; gr[2] += fg[2];
    }
    const float* ps = a.pose_sign + (long)b * a.J * 3;
    for (int k = a.vj_ptr[v]; k < a.vj_ptr[v + 1]; k++) {
      const float w = a.vj_val[k];
      const int j = a.vj_idx[k];
      gr[0] = fmaf(w, ps[j * 3], gr[0]);
      gr[1] = fmaf(w, ps[j * 3 + 1], gr[1]);
      gr[2] = fmaf(w, ps[j * 3 + 2], gr[2]);
    }
    if (a.grad_cam) {
      float* o = a.grad_cam + ((long)b * a.V0 + a.perm[v]) * 3;
      o[0] = gr[0]; o[1] = gr[1]; o[2] = gr[2];
    }
  }
  const float sv = block_sum(l_v, sh);
  if (threadIdx.x == 0) a.partial[0 * a.npart + blockIdx.x] = sv * a.s_vertex;
}

__global__ __launch_bounds__(256) void k_loss_finalize(const float* __restrict__ partial, int npart, int n0, int n1, int n2, int n3,
                                float* __restrict__ losses) {
  const int which = blockIdx.x;
  const int n = which == 0 ? n0 : which == 1 ? n1 : which == 2 ? n2 : n3;
  // 256 threads, four independent loads in flight each (64 threads with one dependent load per iteration took 56 us for a few
  // thousand partials); fixed summation order: deterministic
  __shared__ double wsum[4];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const float* pp = partial + (long)which * npart;
  int i = threadIdx.x;
  for (; i + 768 < n; i += 1024) {
    s0 += (double)pp[i];
    s1 += (double)pp[i + 256];
    s2 += (double)pp[i + 512];
    s3 += (double)pp[i + 768];
  }
  for (; i < n; i += 256) s0 += (double)pp[i];
  double s = (s0 + s1) + (s2 + s3);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) losses[which] = (float)((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
}

// ---- CoordLoss on a small tensor (lib/core/loss.py:10-23; the lifted-pose term of lib/core/base.py:128,139) -------
//   loss = w * mean | pred * valid - target * valid |,   grad = w * sign(pred * valid - target * valid) * valid / n
// One block: a [B, J, 3] pose is a few thousand numbers; fixed summation order (deterministic).
__global__ __launch_bounds__(1024) void k_coord_loss(const float* __restrict__ pred, const float* __restrict__ target,
                                                     const float* __restrict__ valid, int per_mask, long n, float w,
                                                     float* __restrict__ loss, float* __restrict__ grad) {
  __shared__ double wsum[16];
  double s = 0.0;
  const float gscale = w / (float)n;
  for (long i = threadIdx.x; i < n; i += 1024) {
    const float v = valid ? valid[i / per_mask] : 1.f;
    const float d = pred[i] * v - target[i] * v;                // loss.py:19-20: both sides masked, then the difference
    s += (double)fabsf(d);
    if (grad) grad[i] = d > 0.f ? gscale * v : d < 0.f ? -gscale * v : 0.f;     // torch's abs backward: sign(d), 0 at 0
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; k++) t += wsum[k];
    loss[0] = (float)(t / (double)n) * w;
  }
}

// ---- test-step / demo epilogue (lib/core/base.py:200-204, demo/run.py:169-171) ---------------------------------
//   mesh[b, i] = scale * cam_mesh[b, perm[i]]          (tree order incl. fake vertices -> mesh-model vertex order)
//   joints[b, j] = sum_k jr_val[k] * mesh[b, jr_idx[k]]  (CSR row j of the joint regressor)
__global__ __launch_bounds__(256) void k_mesh_epilogue(const float* __restrict__ cam, int V0, const int* __restrict__ perm,
                                                       int nv, float scale, const int* __restrict__ jr_ptr,
                                                       const int* __restrict__ jr_idx, const float* __restrict__ jr_val,
                                                       int J, float* __restrict__ mesh, float* __restrict__ joints, int B) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long nmesh = (long)B * nv;
  if (idx < nmesh) {
    const int b = (int)(idx / nv), v = (int)(idx - (long)b * nv);
    const float* p = cam + ((long)b * V0 + perm[v]) * 3;
    if (mesh) {
      float* o = mesh + idx * 3;
      o[0] = p[0] * scale; o[1] = p[1] * scale; o[2] = p[2] * scale;
    }
  } else if (joints != nullptr && idx < nmesh + (long)B * J) {
    const long q = idx - nmesh;
    const int b = (int)(q / J), j = (int)(q - (long)b * J);
    const float* cb = cam + (long)b * V0 * 3;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int k = jr_ptr[j]; k < jr_ptr[j + 1]; k++) {
      const float w = jr_val[k];
      const float* p = cb + (long)perm[jr_idx[k]] * 3;
      s0 = fmaf(w, p[0] * scale, s0);
      s1 = fmaf(w, p[1] * scale, s1);
      s2 = fmaf(w, p[2] * scale, s2);
    }
    float* o = joints + q * 3;
    o[0] = s0; o[1] = s1; o[2] = s2;
  }
}

}  // namespace p2m

using namespace p2m;

extern "C" int p2m_mesh_epilogue(const float* cam_mesh, int32_t V0, const int32_t* perm, int32_t nv, float scale,
                                 const int32_t* jr_ptr, const int32_t* jr_idx, const float* jr_val, int32_t J,
                                 float* mesh, float* joints, int32_t B, void* stream) {
  P2M_CHECK_ARG(cam_mesh && perm && (mesh || joints), "null pointer");
  P2M_CHECK_ARG(joints == nullptr || (jr_ptr && jr_idx && jr_val && J > 0), "joints requested without a regressor");
  P2M_CHECK_ARG(V0 > 0 && nv > 0, "empty shape");
  if (B <= 0) return P2M_OK;
  const long tot = (long)B * nv + (joints ? (long)B * J : 0);
  hipLaunchKernelGGL(k_mesh_epilogue, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, cam_mesh, V0, perm, nv,
                     scale, jr_ptr, jr_idx, jr_val, J, mesh, joints, B);
  return check_launch("mesh_epilogue");
}

static long loss_nparts(int B, int nv, int F, int J) {
  const long nb_face = cdiv((long)B * F, 256), nb_vert = cdiv((long)B * nv, 256), nb_pose = cdiv((long)B * J, 256);
  long npart = nb_face > nb_vert ? nb_face : nb_vert;
  return nb_pose > npart ? nb_pose : npart;
}

extern "C" int64_t p2m_mesh_loss_workspace(int32_t B, int32_t nv, int32_t F, int32_t J) {
  return (long)B * F * 9 + (long)B * J * 3 + 4 * loss_nparts(B, nv, F, J);    // floats
}

extern "C" int p2m_mesh_loss(const float* cam_mesh, int32_t V0, const int32_t* perm, int32_t nv, const float* gt_mesh,
                             const float* valid_mesh, const int32_t* faces, int32_t F, const int32_t* vf_ptr,
                             const int32_t* vf_idx, const int32_t* jr_ptr, const int32_t* jr_idx, const float* jr_val,
                             const int32_t* vj_ptr, const int32_t* vj_idx, const float* vj_val, int32_t J,
                             const float* gt_pose, const float* valid_pose, float w_vertex, float w_normal,
                             float w_edge, float w_joint, float* workspace, float* losses, float* grad_cam, int32_t B,
                             void* stream) {
  P2M_CHECK_ARG(cam_mesh && perm && gt_mesh && faces && vf_ptr && vf_idx && gt_pose && workspace && losses,
                "null pointer");
  P2M_CHECK_ARG(jr_ptr && jr_idx && jr_val && vj_ptr && vj_idx && vj_val, "null joint-regressor CSR/CSC");
  P2M_CHECK_ARG(B > 0 && V0 > 0 && nv > 0 && F > 0 && J > 0, "empty shape");
  LossArgs a;
  a.cam = cam_mesh; a.perm = perm; a.gt_mesh = gt_mesh; a.valid_mesh = valid_mesh; a.faces = faces;
  a.vf_ptr = vf_ptr; a.vf_idx = vf_idx; a.gt_pose = gt_pose; a.valid_pose = valid_pose;
  a.jr_ptr = jr_ptr; a.jr_idx = jr_idx; a.jr_val = jr_val; a.vj_ptr = vj_ptr; a.vj_idx = vj_idx; a.vj_val = vj_val;
  a.B = B; a.V0 = V0; a.nv = nv; a.F = F; a.J = J;
  const int nb_face = cdiv((long)B * F, 256), nb_vert = cdiv((long)B * nv, 256), nb_pose = cdiv((long)B * J, 256);
  a.npart = (int)loss_nparts(B, nv, F, J);
  a.face_grad = workspace;
  a.pose_sign = workspace + (long)B * F * 9;
  a.partial = a.pose_sign + (long)B * J * 3;
  a.grad_cam = grad_cam;
  a.s_vertex = w_vertex / ((float)B * nv * 3);
  a.s_normal = w_normal / ((float)B * F * 3);
  a.s_edge = w_edge / ((float)B * F * 3);
  a.s_joint = w_joint / ((float)B * J * 3);
  hipStream_t s = (hipStream_t)stream;
  if (grad_cam) {
    hipError_t e = hipMemsetAsync(grad_cam, 0, sizeof(float) * (size_t)B * V0 * 3, s);   // fake vertices get no gradient
    if (e != hipSuccess) {
      set_error("p2m_mesh_loss: memset failed: %s", hipGetErrorString(e));
      return P2M_ERR_HIP;
    }
  }
  hipLaunchKernelGGL(k_pose_regress, dim3(nb_pose), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_face_terms, dim3(nb_face), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_vertex_grad, dim3(nb_vert), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_loss_finalize, dim3(4), dim3(256), 0, s, a.partial, a.npart, nb_vert, nb_face, nb_face, nb_pose, losses);
  return check_launch("mesh_loss");
}

extern "C" int p2m_coord_loss(const float* pred, const float* target, const float* valid, int32_t per_mask, int64_t n,
                              float w, float* loss, float* grad, void* stream) {
  P2M_CHECK_ARG(pred && target && loss && n > 0 && per_mask > 0, "null pointer or empty shape");
  hipLaunchKernelGGL(k_coord_loss, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, target, valid, per_mask, (long)n, w,
                     loss, grad);
  return check_launch("coord_loss");
}
