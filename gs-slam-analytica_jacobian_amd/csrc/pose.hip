// pose.hip -- one tracking pose step on the device: Adam on (cam_trans_delta, cam_rot_delta, exposure a, b) and
// update_pose, then the camera matrices the next render needs (gfx950).  SURVEY 8(f)-2.
//
// Semantics: torch.optim.Adam (default: no weight decay / amsgrad) over the parameter groups slam_frontend.py:135-160
// sets up, whose values are zero before every step (update_pose resets the deltas, pose_utils.py:91-92), so the step IS
// tau = [rho, theta]; then pose_utils.py:61-93: W2C <- SE3_exp(tau) * W2C with the small-angle branches at 1e-5, and
// converged = |tau| < threshold.  Derived outputs: world_view_transform = W2C^T, full_proj_transform =
// W2C^T * projection_matrix, camera_center = -R^-1 t (camera_utils.py:95-109).  The reference does this with ~60 tiny
// torch kernels and one host read-back per iteration; here it is one single-lane kernel and no read-back.
#include "gsaj_common.h"

// pose_state layout (floats): see include/gsaj.h
#define PS_W2C 0
#define PS_M 16
#define PS_V 24
#define PS_STEP 32
#define PS_EXP 33
#define PS_VIEW 35
#define PS_PROJ 51
#define PS_CAMPOS 67
#define PS_TAU 70
#define PS_NORM 76
#define PS_CONV 77

struct PoseStepParams {
  const float *g_tau, *g_exp, *projection;
  float lr[8];
  float beta1, beta2, eps, threshold;
  float *st;
  const uint8_t *active;  // batched launch: per-pose enable flags (NULL: all)
  const uint32_t *skip;   // (may be NULL) a non-zero word makes the step a no-op: the frame's abort flag (gsaj_forward_abort_flag),
  size_t skip_stride;     //   pose k's word `skip_stride` bytes after pose k - 1's
};

__device__ void so3_exp_and_v(const float *th, float R[9], float V[9]) {
  const float W[9] = {0.f, -th[2], th[1], th[2], 0.f, -th[0], -th[1], th[0], 0.f};
  float W2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) W2[i * 3 + j] = W[i * 3 + 0] * W[0 * 3 + j] + W[i * 3 + 1] * W[1 * 3 + j] + W[i * 3 + 2] * W[2 * 3 + j];
  const float angle = sqrtf(th[0] * th[0] + th[1] * th[1] + th[2] * th[2]);
  float a, b, c, d;  // R = I + a W + b W2;  V = I + c W + d W2
  if (angle < 1e-5f) {
    a = 1.f; b = 0.5f; c = 0.5f; d = 1.0f / 6.0f;
  } else {
    const float s = sinf(angle), co = cosf(angle), a2 = angle * angle;
    a = s / angle;
    b = (1.f - co) / a2;
    c = b;
    d = (angle - s) / (a2 * angle);
  }
  for (int i = 0; i < 9; i++) {
    const float id = (i % 4 == 0) ? 1.f : 0.f;
    R[i] = id + a * W[i] + b * W2[i];
    V[i] = id + c * W[i] + d * W2[i];
  }
}

__global__ void k_pose_adam_step(PoseStepParams p) {
  if (threadIdx.x != 0) return;
  // batched launch: workgroup k steps pose k ([K,6] gradients, [K,2] exposure gradients, K states); `active` (may be NULL) masks
  // keyframes whose pose stays fixed (the reference skips uid 0, utils/slam_backend.py:255-258)
  if (p.active && !p.active[blockIdx.x]) return;
  // the frame these gradients belong to was aborted on the device (its kernels returned at once: dL/dtau and the loss terms are
  // the PREVIOUS iteration's): Adam must not step on them, nor advance its moments
  if (p.skip && *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(p.skip) + p.skip_stride * blockIdx.x) != 0u) return;
  p.g_tau += 6 * (size_t)blockIdx.x;
  if (p.g_exp) p.g_exp += 2 * (size_t)blockIdx.x;
  float *st = p.st + (size_t)GSAJ_POSE_STATE_FLOATS * blockIdx.x;
  const float t = st[PS_STEP] + 1.f;
  st[PS_STEP] = t;
  const float bc1 = 1.f - powf(p.beta1, t), bc2 = 1.f - powf(p.beta2, t);
  float delta[8];
  for (int i = 0; i < 8; i++) {
    const float g = (i < 6) ? p.g_tau[i] : (p.g_exp ? p.g_exp[i - 6] : 0.f);
    const float m = p.beta1 * st[PS_M + i] + (1.f - p.beta1) * g;
    const float v = p.beta2 * st[PS_V + i] + (1.f - p.beta2) * g * g;
    st[PS_M + i] = m;
    st[PS_V + i] = v;
    const float step_size = p.lr[i] / bc1;
    const float denom = sqrtf(v) / sqrtf(bc2) + p.eps;
    delta[i] = -step_size * (m / denom);
  }
  st[PS_EXP + 0] += delta[6];
  st[PS_EXP + 1] += delta[7];
  float R[9], V[9];
  so3_exp_and_v(delta + 3, R, V);
  const float tr[3] = {V[0] * delta[0] + V[1] * delta[1] + V[2] * delta[2], V[3] * delta[0] + V[4] * delta[1] + V[5] * delta[2],
                       V[6] * delta[0] + V[7] * delta[1] + V[8] * delta[2]};
  float w[16], nw[16];
  for (int i = 0; i < 16; i++) w[i] = st[PS_W2C + i];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 4; j++)
      nw[i * 4 + j] = R[i * 3 + 0] * w[0 * 4 + j] + R[i * 3 + 1] * w[1 * 4 + j] + R[i * 3 + 2] * w[2 * 4 + j] + tr[i] * w[3 * 4 + j];
  }
  nw[12] = 0.f; nw[13] = 0.f; nw[14] = 0.f; nw[15] = 1.f;
  for (int i = 0; i < 16; i++) st[PS_W2C + i] = nw[i];
  // world_view_transform = W2C^T (row-major), full_proj_transform = W2C^T * projection_matrix, camera centre = -R^T t
  float vw[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) vw[i * 4 + j] = nw[j * 4 + i];
  for (int i = 0; i < 16; i++) st[PS_VIEW + i] = vw[i];
  if (p.projection) {
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        float s = 0.f;
        for (int k = 0; k < 4; k++) s += vw[i * 4 + k] * p.projection[k * 4 + j];
        st[PS_PROJ + i * 4 + j] = s;
      }
  }
  {  // camera_center = inverse(world_view_transform)[3, :3] = -R^-1 t (general 3x3 inverse: W2C may carry a scale)
    const float a = nw[0], b = nw[1], c = nw[2], d = nw[4], e = nw[5], f = nw[6], g = nw[8], h = nw[9], i = nw[10];
    const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const float idet = 1.f / (a * A + b * B + c * C);
    const float inv[9] = {A * idet, -(b * i - c * h) * idet, (b * f - c * e) * idet, B * idet, (a * i - c * g) * idet, -(a * f - c * d) * idet,
                          C * idet, -(a * h - b * g) * idet, (a * e - b * d) * idet};
    for (int j = 0; j < 3; j++) st[PS_CAMPOS + j] = -(inv[j * 3 + 0] * nw[3] + inv[j * 3 + 1] * nw[7] + inv[j * 3 + 2] * nw[11]);
  }
  float n2 = 0.f;
  for (int i = 0; i < 6; i++) {
    st[PS_TAU + i] = delta[i];
    n2 += delta[i] * delta[i];
  }
  const float nrm = sqrtf(n2);
  st[PS_NORM] = nrm;
  st[PS_CONV] = (nrm < p.threshold) ? 1.f : 0.f;
}

extern "C" int gsaj_pose_state_floats(void) { return GSAJ_POSE_STATE_FLOATS; }

extern "C" int gsaj_pose_adam_step(const float *dL_dtau, const float *dL_dexposure, float lr_rot, float lr_trans, float lr_exp_a,
                                   float lr_exp_b, float beta1, float beta2, float eps, float converged_threshold,
                                   const float *projection_matrix, float *pose_state, const uint32_t *skip, void *stream) {
  if (!dL_dtau || !pose_state) {
    gsaj_set_error("gsaj_pose_adam_step: dL_dtau and pose_state are required");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  PoseStepParams p;
  p.g_tau = dL_dtau; p.g_exp = dL_dexposure; p.projection = projection_matrix;
  for (int i = 0; i < 3; i++) { p.lr[i] = lr_trans; p.lr[3 + i] = lr_rot; }
  p.lr[6] = lr_exp_a; p.lr[7] = lr_exp_b;
  p.beta1 = beta1; p.beta2 = beta2; p.eps = eps; p.threshold = converged_threshold;
  p.st = pose_state;
  p.active = nullptr;
  p.skip = skip; p.skip_stride = 0;
  hipLaunchKernelGGL(k_pose_adam_step, dim3(1), dim3(64), 0, (hipStream_t)stream, p);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

extern "C" int gsaj_pose_adam_step_batch(int K, const float *dL_dtau, const float *dL_dexposure, const uint8_t *active, float lr_rot,
                                         float lr_trans, float lr_exp_a, float lr_exp_b, float beta1, float beta2, float eps,
                                         float converged_threshold, const float *projection_matrix, float *pose_states,
                                         const uint32_t *skip, size_t skip_stride_bytes, void *stream) {
  if (K <= 0 || !dL_dtau || !pose_states) {
    gsaj_set_error("gsaj_pose_adam_step_batch: K > 0, dL_dtau and pose_states are required");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  PoseStepParams p;
  p.g_tau = dL_dtau; p.g_exp = dL_dexposure; p.projection = projection_matrix;
  for (int i = 0; i < 3; i++) { p.lr[i] = lr_trans; p.lr[3 + i] = lr_rot; }
  p.lr[6] = lr_exp_a; p.lr[7] = lr_exp_b;
  p.beta1 = beta1; p.beta2 = beta2; p.eps = eps; p.threshold = converged_threshold;
  p.st = pose_states;
  p.active = active;
  p.skip = skip; p.skip_stride = skip_stride_bytes;
  hipLaunchKernelGGL(k_pose_adam_step, dim3((unsigned)K), dim3(64), 0, (hipStream_t)stream, p);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}
