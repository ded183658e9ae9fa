/* c_abi_example.c -- the Gram engine used from plain C through include/qkgram.h (no Python, no torch).
 *
 *   gcc -O2 -I include examples/c_abi_example.c -o /tmp/qk_example \
 *       -L qml-cutensornet_amd -lqkgram -Wl,-rpath,$PWD/qml-cutensornet_amd -lm
 *
 * Builds three 4-qubit product states |psi(t)> = prod_k (cos t_k |0> + i sin t_k |1>) as bond-1 MPS, asks for their
 * Gram matrix K[j][i] = |<psi_i|psi_j>|^2 and checks it against the closed form prod_k cos^2(t_ik - t_jk).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "qkgram.h"

#define N_SITES 4
#define N_STATES 3

int main(void) {
  static const double angle[N_STATES][N_SITES] = {{0.1, 0.7, 1.3, 0.4}, {0.9, 0.2, 0.5, 1.1}, {0.3, 0.3, 0.8, 0.6}};
  /* site tensors: complex128 (re, im interleaved) of shape (1, 2, 1), layout QK_LAYOUT_LPR */
  static double tensor[N_STATES][N_SITES][4];
  const double* ptrs[N_STATES * N_SITES];
  int32_t dims[N_STATES][N_SITES + 1];
  for (int s = 0; s < N_STATES; ++s) {
    for (int k = 0; k <= N_SITES; ++k) dims[s][k] = 1;
    for (int k = 0; k < N_SITES; ++k) {
      tensor[s][k][0] = cos(angle[s][k]), tensor[s][k][1] = 0.0; /* amplitude of |0> */
      tensor[s][k][2] = 0.0, tensor[s][k][3] = sin(angle[s][k]); /* amplitude of |1> = i sin */
      ptrs[s * N_SITES + k] = tensor[s][k];
    }
  }
  qk_ctx* ctx = NULL;
  if (qk_ctx_create(0, &ctx) != QK_OK) {
    fprintf(stderr, "no usable gfx950 device: %s\n", qk_last_error());
    return 2; /* there is no CPU fallback */
  }
  qk_mps_set* set = NULL;
  if (qk_mps_set_create(ctx, N_STATES, N_SITES, &dims[0][0], ptrs, QK_LAYOUT_LPR, &set) != QK_OK) {
    fprintf(stderr, "qk_mps_set_create: %s\n", qk_last_error());
    return 1;
  }
  double K[N_STATES][N_STATES];
  if (qk_gram_host(ctx, set, NULL, &K[0][0], N_STATES) != QK_OK) {
    fprintf(stderr, "qk_gram_host: %s\n", qk_last_error());
    return 1;
  }
  double worst = 0.0;
  for (int j = 0; j < N_STATES; ++j)
    for (int i = 0; i < N_STATES; ++i) {
      double ref = 1.0;
      for (int k = 0; k < N_SITES; ++k) ref *= cos(angle[i][k] - angle[j][k]) * cos(angle[i][k] - angle[j][k]);
      worst = fmax(worst, fabs(K[j][i] - ref));
    }
  printf("K[1][0] = %.15f, max |K - closed form| = %.3e\n", K[1][0], worst);
  qk_mps_set_destroy(set);
  qk_ctx_destroy(ctx);
  if (!(worst < 1e-12)) return 1;

  /* The same Gram through the multi-GPU entry points: every device of the node gets a share of the states, ONE all-gather
   * of the packed images gives every device the whole set, every device sweeps its share of the pairs and ONE
   * ncclAllGather (RCCL over xGMI) joins the values.  On a one-GPU box this is a communicator of one rank. */
  int n_dev = qk_device_count();
  if (n_dev > N_STATES) n_dev = N_STATES;
  qk_comm* comm = NULL;
  if (qk_comm_init_all(n_dev, NULL, &comm) != QK_OK) {
    fprintf(stderr, "qk_comm_init_all: %s\n", qk_last_error());
    return 1;
  }
  qk_mps_set* share[16] = {0};
  qk_mps_set* full[16] = {0};
  int32_t lo[16];
  const int per = (N_STATES + n_dev - 1) / n_dev;
  for (int r = 0; r < n_dev; ++r) {
    lo[r] = r * per < N_STATES ? r * per : N_STATES;
    const int cnt = (lo[r] + per <= N_STATES ? per : N_STATES - lo[r]);
    if (cnt > 0 && qk_mps_set_create(qk_comm_ctx(comm, r), cnt, N_SITES, &dims[lo[r]][0], ptrs + lo[r] * N_SITES, QK_LAYOUT_LPR, &share[r]) != QK_OK) {
      fprintf(stderr, "qk_mps_set_create (rank %d): %s\n", r, qk_last_error());
      return 1;
    }
  }
  if (qk_mps_set_allgather(comm, share, lo, N_STATES, full) != QK_OK || qk_gram_sharded(comm, full, NULL, &K[0][0], N_STATES) != QK_OK) {
    fprintf(stderr, "sharded gram: %s\n", qk_last_error());
    return 1;
  }
  double worst_sharded = 0.0;
  for (int j = 0; j < N_STATES; ++j)
    for (int i = 0; i < N_STATES; ++i) {
      double ref = 1.0;
      for (int k = 0; k < N_SITES; ++k) ref *= cos(angle[i][k] - angle[j][k]) * cos(angle[i][k] - angle[j][k]);
      worst_sharded = fmax(worst_sharded, fabs(K[j][i] - ref));
    }
  printf("sharded over %d device(s): max |K - closed form| = %.3e\n", n_dev, worst_sharded);
  for (int r = 0; r < n_dev; ++r) {
    qk_mps_set_destroy(share[r]);
    qk_mps_set_destroy(full[r]);
  }
  qk_comm_destroy(comm);
  return worst_sharded < 1e-12 ? 0 : 1;
}
