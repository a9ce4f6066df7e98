// Stand-alone check and timing of the single-launch dense Cholesky solve (sfmba_chol.hip):
// random SPD system, result against a host factorisation, time per launch, and the phase time
// stamps of every block row.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 chol_flow_test.hip -o chol_flow_test
#include "../../structure_from_motion_amd/csrc/sfmba_chol.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n_real = (argc > 1 ? atoi(argv[1]) : 282) / 6 * 6;      // 6 x 6 blocks, every block present
    const int n = (n_real + 31) / 32 * 32, nb = n / 32;
    std::vector<double> A((size_t)n * n, 0.0), b(n_real), x(n_real);
    srand(1);
    std::vector<double> G((size_t)n_real * n_real);
    for (auto& g : G) g = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n_real; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = i == j ? 1e-3 : 0.0;
            for (int k = 0; k < n_real; ++k) s += G[(size_t)i * n_real + k] * G[(size_t)j * n_real + k];
            A[(size_t)i * n + j] = A[(size_t)j * n + i] = s;
        }
    for (auto& v : b) v = rand() / (double)RAND_MAX - 0.5;
    // host solve
    std::vector<double> L(A), y(n_real), xr(n_real);
    for (int j = 0; j < n_real; ++j) {
        double d = L[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        d = std::sqrt(d);
        L[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n_real; ++i) {
            double s = L[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n_real; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * y[k]; y[i] = s / L[(size_t)i * n + i]; }
    for (int i = n_real - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n_real; ++k) s -= L[(size_t)k * n + i] * xr[k]; xr[i] = s / L[(size_t)i * n + i]; }

    double *dA, *db, *dx, *dbuf; int* dfail; long long* dtrace;
    CHECK(hipMalloc(&dA, A.size() * 8)); CHECK(hipMalloc(&db, n_real * 8)); CHECK(hipMalloc(&dx, n_real * 8));
    CHECK(hipMalloc(&dbuf, sfmba::chol_flow_words(n) * 8)); CHECK(hipMalloc(&dfail, 4)); CHECK(hipMalloc(&dtrace, 2 * nb * 8 * 8));
    const int n_cam = n_real / 6;
    std::vector<int> ptr(n_cam + 1), colv((size_t)n_cam * n_cam);
    std::vector<double> S((size_t)n_cam * n_cam * 36);
    for (int r = 0; r <= n_cam; ++r) ptr[r] = r * n_cam;
    for (int r = 0; r < n_cam; ++r)
        for (int c = 0; c < n_cam; ++c) {
            colv[(size_t)r * n_cam + c] = c;
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) S[((size_t)r * n_cam + c) * 36 + 6 * i + j] = A[(size_t)(6 * r + i) * n + 6 * c + j];
        }
    int *dptr, *dcol; double* dS;
    CHECK(hipMalloc(&dptr, ptr.size() * 4)); CHECK(hipMalloc(&dcol, colv.size() * 4)); CHECK(hipMalloc(&dS, S.size() * 8));
    CHECK(hipMemcpy(dptr, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dcol, colv.data(), colv.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice));
    sfmba::Dev dv{};
    dv.n_cam = n_cam; dv.model = sfmba::MODEL_PINHOLE; dv.bsr_ptr = dptr; dv.bsr_col = dcol; dv.Sfull = dS;
    CHECK(hipMemcpy(db, b.data(), n_real * 8, hipMemcpyHostToDevice));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 20;
    for (int it = 0; it < 3; ++it) CHECK(sfmba::launch_chol_flow(dv, dA, n, n_real, db, dx, dbuf, dfail, nullptr, nullptr, st, dtrace));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int it = 0; it < reps; ++it) CHECK(sfmba::launch_chol_flow(dv, dA, n, n_real, db, dx, dbuf, dfail, nullptr, nullptr, st, nullptr));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    int fail; CHECK(hipMemcpy(&fail, dfail, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(x.data(), dx, n_real * 8, hipMemcpyDeviceToHost));
    double err = 0, nrm = 0;
    for (int i = 0; i < n_real; ++i) { err = std::fmax(err, std::fabs(x[i] - xr[i])); nrm = std::fmax(nrm, std::fabs(xr[i])); }
    printf("n %d (%d block rows): %.1f us per solve (prepare + kernel), fail %d, max |x - x_host| / max |x_host| = %.2e\n", n_real, nb, 1e3 * ms / reps, fail, err / nrm);
    CHECK(sfmba::launch_chol_flow(dv, dA, n, n_real, db, dx, dbuf, dfail, nullptr, nullptr, st, dtrace));
    CHECK(hipStreamSynchronize(st));
    std::vector<long long> tr((size_t)nb * 16);
    CHECK(hipMemcpy(tr.data(), dtrace, tr.size() * 8, hipMemcpyDeviceToHost));
    const long long t0 = tr[0];
    printf("row: start  tiles_done  factored  inverted(+barrier)  published  back_wait_done  end   [us since row 0 started]\n");
    for (int i = 0; i < nb; ++i) {
        printf("%3d:", i);
        for (int k = 0; k < 7; ++k) printf(" %8.2f", (tr[8 * i + k] - t0) * 0.01);
        printf("\n");
    }
    printf("shader clock over the factorisation of row 0: %lld cycles in %.2f us = %.0f MHz\n", tr[8 * nb + 2] - tr[8 * nb + 1],
           (tr[2] - tr[1]) * 0.01, (tr[8 * nb + 2] - tr[8 * nb + 1]) / ((tr[2] - tr[1]) * 0.01));
    return 0;
}
