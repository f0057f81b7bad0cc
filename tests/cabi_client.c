/* A plain C client of the drop-in boundary (include/inflx_hip.h) -- TEST INFRASTRUCTURE.
 * What a Rust/C maintainer of the reference would write: open an artefact, call the reference-shaped entry
 * points, read the error text on failure.  No Python, no torch in this process.
 *   usage: cabi_client ARTEFACT N0 N1 x0a x0b x1a x1b OUT.bin p0 [p1 ...]
 * writes N0*N1*6 doubles (complete_analysis) followed by N0*N1 doubles (consistency_only) to OUT.bin.
 * Exit codes the tests look at: 3 = inflx_open failed (status in the message), 4 = the artefact does not have two fields /
 * the given number of parameters and inflx_complete_analysis refused it with INFLX_ERR_SHAPE (anything else: 9). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "inflx_hip.h"

int main(int argc, char** argv) {
  if (argc < 10) {
    fprintf(stderr, "usage: %s ARTEFACT N0 N1 x0a x0b x1a x1b OUT.bin p0 [p1 ...]\n", argv[0]);
    return 64;
  }
  const size_t n0 = (size_t)atol(argv[2]), n1 = (size_t)atol(argv[3]);
  const double start_stop[4] = {atof(argv[4]), atof(argv[5]), atof(argv[6]), atof(argv[7])};
  const size_t n_p = (size_t)(argc - 9);
  double* p = malloc(n_p * sizeof(double));
  for (size_t k = 0; k < n_p; ++k) p[k] = atof(argv[9 + k]);

  int ndev = 0;
  if (inflx_device_count(&ndev) != INFLX_OK || ndev < 1) {
    fprintf(stderr, "no device: %s\n", inflx_last_error());
    return 2;
  }
  inflx_model* model = NULL;
  int rc = inflx_open(argv[1], 0, &model);
  if (rc != INFLX_OK) {
    fprintf(stderr, "inflx_open failed (%d): %s\n", rc, inflx_last_error());
    return 3;
  }
  if (inflx_n_fields(model) != 2 || inflx_n_parameters(model) != n_p) {
    fprintf(stderr, "model %s: %u fields, %u parameters (got %zu)\n", inflx_model_name(model), inflx_n_fields(model), inflx_n_parameters(model), n_p);
    /* the sweep itself must refuse such an artefact: Hesse2D::new asserts n_fields == 2 (src/hesse_bindings.rs:203) */
    double probe[6];
    rc = inflx_complete_analysis(model, p, inflx_n_parameters(model), probe, start_stop, 1, 1, 0, 0);
    fprintf(stderr, "inflx_complete_analysis on it: status %d: %s\n", rc, inflx_last_error());
    return rc == INFLX_ERR_SHAPE ? 4 : 9;
  }
  /* a wrong parameter count must be refused with the shape status, and leave the handle usable */
  double* six = calloc(n0 * n1 * 6, sizeof(double));
  rc = inflx_complete_analysis(model, p, n_p + 1, six, start_stop, n0, n1, 0, 0);
  if (rc != INFLX_ERR_SHAPE) {
    fprintf(stderr, "expected INFLX_ERR_SHAPE for a wrong parameter count, got %d\n", rc);
    return 5;
  }
  rc = inflx_complete_analysis(model, p, n_p, six, start_stop, n0, n1, 0, 0);
  if (rc != INFLX_OK) {
    fprintf(stderr, "inflx_complete_analysis failed (%d): %s\n", rc, inflx_last_error());
    return 6;
  }
  double* one = calloc(n0 * n1, sizeof(double));
  rc = inflx_consistency_only(model, p, n_p, one, start_stop, n0, n1, 0, 0);
  if (rc != INFLX_OK) {
    fprintf(stderr, "inflx_consistency_only failed (%d): %s\n", rc, inflx_last_error());
    return 7;
  }
  /* one call, several GPUs: the artefact on two handles (device 0 twice -- a one-GPU box), the reference-shaped entry
   * point with threads = 0 ("the whole machine", src/anguelova.rs:524-540); the slabs the two devices copy into `six2`
   * must be the single-device result bit for bit, and the partition must be what inflx_shard_plan says */
  {
    const int devices[2] = {0, 0};
    inflx_multi* multi = NULL;
    rc = inflx_open_multi(argv[1], devices, 2, &multi);
    if (rc != INFLX_OK || inflx_multi_device_count(multi) != 2) {
      fprintf(stderr, "inflx_open_multi failed (%d): %s\n", rc, inflx_last_error());
      return 10;
    }
    double* six2 = calloc(n0 * n1 * 6, sizeof(double));
    rc = inflx_complete_analysis_multi(multi, p, n_p, six2, start_stop, n0, n1, 0, 0);
    if (rc != INFLX_OK || memcmp(six, six2, n0 * n1 * 6 * sizeof(double)) != 0) {
      fprintf(stderr, "inflx_complete_analysis_multi: status %d (%s) or a result that differs from the single-device one\n", rc, inflx_last_error());
      return 11;
    }
    size_t plan[5];
    if (inflx_shard_plan(1, n0, 2, 1, plan) != INFLX_OK || plan[0] != 1 || plan[3] != (n0 + 1) / 2 || plan[3] + plan[4] != n0) {
      fprintf(stderr, "inflx_shard_plan: unexpected partition of %zu rows\n", n0);
      return 12;
    }
    rc = inflx_complete_analysis_multi(multi, p, n_p + 1, six2, start_stop, n0, n1, 0, 0);
    if (rc != INFLX_ERR_SHAPE) return 13;
    inflx_summary whole, parts;
    if (inflx_sweep_device_stats(model, p, 1, n_p, NULL, 0, start_stop, n0, n1, 0, n0, NULL, &whole) != INFLX_OK ||
        inflx_sweep_stats_multi(multi, p, 1, n_p, start_stop, n0, n1, 0, &parts) != INFLX_OK || memcmp(&whole, &parts, sizeof whole) != 0) {
      fprintf(stderr, "inflx_sweep_stats_multi differs from inflx_sweep_device_stats: %s\n", inflx_last_error());
      return 14;
    }
    inflx_close_multi(multi);
    free(six2);
  }
  /* a plane subset of the planes-layout result (what calc_V_array / calc_H_array copy): planes 1-2 of the six = planes 1-2 of the
   * full planes sweep, and plane k of the planes sweep = value k of the record sweep above */
  {
    double* planes = (double*)malloc(n0 * n1 * 6 * sizeof(double));
    double* two = (double*)malloc(n0 * n1 * 2 * sizeof(double));
    rc = inflx_sweep_host(model, INFLX_SWEEP_COMPLETE, p, 1, n_p, planes, start_stop, n0, n1, 0, n0, INFLX_SOA);
    if (rc == INFLX_OK) rc = inflx_sweep_host_planes(model, INFLX_SWEEP_COMPLETE, p, 1, n_p, two, start_stop, n0, n1, 0, n0, 1, 2);
    int same = rc == INFLX_OK && memcmp(two, planes + n0 * n1, n0 * n1 * 2 * sizeof(double)) == 0;
    for (size_t i = 0; same && i < n0 * n1; ++i) same = memcmp(&planes[2 * n0 * n1 + i], &six[6 * i + 2], sizeof(double)) == 0;
    if (!same) {
      fprintf(stderr, "inflx_sweep_host_planes: status %d (%s) or planes that differ from the full sweep\n", rc, inflx_last_error());
      return 12;
    }
    if (inflx_sweep_host_planes(model, INFLX_SWEEP_COMPLETE, p, 1, n_p, two, start_stop, n0, n1, 0, n0, 5, 2) == INFLX_OK) return 13; /* planes 5-6 of six */
    free(planes);
    free(two);
  }
  FILE* f = fopen(argv[8], "wb");
  if (!f) return 8;
  fwrite(six, sizeof(double), n0 * n1 * 6, f);
  fwrite(one, sizeof(double), n0 * n1, f);
  fclose(f);
  inflx_close(model);
  free(six);
  free(one);
  free(p);
  printf("ok %s\n", argv[1]);
  return 0;
}
