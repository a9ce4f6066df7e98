/* A caller of libsfmba.so written in plain C: reads a problem in the five-array layout of
 * optimize_data.npz (/root/reference/bundle_adjustment.py:125-126) from a flat binary file,
 * solves it through the C ABI and prints the statistics.  Built and run by tests/test_c_client.py.
 *
 * file layout (little endian): int64 n_cameras, n_points, n_obs; double K[9];
 * double camera_params[n_cameras*12]; double points_3d[n_points*3]; int64 camera_indices[n_obs];
 * int64 point_indices[n_obs]; double points_2d[n_obs*2] */
#include <stdio.h>
#include <stdlib.h>

#include "sfmba.h"

static void* must_read(FILE* f, size_t size, size_t count) {
    void* p = malloc(size * (count ? count : 1));
    if (!p || fread(p, size, count, f) != count) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: client problem.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int64_t* hdr = must_read(f, sizeof(int64_t), 3);
    const int64_t n_cam = hdr[0], n_pts = hdr[1], n_obs = hdr[2];
    double* K = must_read(f, sizeof(double), 9);
    double* cams = must_read(f, sizeof(double), (size_t)n_cam * 12);
    double* pts = must_read(f, sizeof(double), (size_t)n_pts * 3);
    int64_t* ci = must_read(f, sizeof(int64_t), (size_t)n_obs);
    int64_t* pi = must_read(f, sizeof(int64_t), (size_t)n_obs);
    double* uv = must_read(f, sizeof(double), (size_t)n_obs * 2);
    fclose(f);

    sfmba_options opt;
    sfmba_default_options(&opt);
    sfmba_handle* h = NULL;
    int rc = sfmba_create(&opt, &h);
    if (rc != SFMBA_OK) { fprintf(stderr, "sfmba_create: %d\n", rc); return 1; }
    rc = sfmba_set_problem(h, n_cam, n_pts, n_obs, cams, pts, ci, pi, uv, K);
    if (rc != SFMBA_OK) { fprintf(stderr, "sfmba_set_problem: %d %s\n", rc, sfmba_last_error(h)); return 1; }
    double* res = malloc(sizeof(double) * 2 * (size_t)(n_obs ? n_obs : 1));
    rc = sfmba_residuals(h, res);
    if (rc != SFMBA_OK) { fprintf(stderr, "sfmba_residuals: %d %s\n", rc, sfmba_last_error(h)); return 1; }
    double c0 = 0.0;
    for (int64_t i = 0; i < 2 * n_obs; ++i) c0 += res[i] * res[i];
    sfmba_stats st;
    rc = sfmba_solve(h, &st);
    if (rc != SFMBA_OK) { fprintf(stderr, "sfmba_solve: %d %s\n", rc, sfmba_last_error(h)); return 1; }
    rc = sfmba_get_params(h, cams, pts);
    if (rc != SFMBA_OK) { fprintf(stderr, "sfmba_get_params: %d %s\n", rc, sfmba_last_error(h)); return 1; }
    printf("status %d iterations %d nfev %d\n", (int)st.status, (int)st.iterations, (int)st.nfev);
    printf("cost_from_residuals %.17g\ninitial_cost %.17g\nfinal_cost %.17g\n", 0.5 * c0, st.initial_cost, st.final_cost);
    printf("camera0 %.17g %.17g %.17g\npoint0 %.17g %.17g %.17g\n", cams[9], cams[10], cams[11], pts[0], pts[1], pts[2]);
    /* an index out of range comes back as an error code with a message, never as a crash */
    ci[0] = n_cam;
    rc = sfmba_set_problem(h, n_cam, n_pts, n_obs, cams, pts, ci, pi, uv, K);
    printf("bad_index_rc %d\nbad_index_msg %s\n", rc, sfmba_last_error(h));
    sfmba_destroy(h);
    return 0;
}
