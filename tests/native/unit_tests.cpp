// tests/native/unit_tests.cpp -- native harness, the counterpart of the reference's
// PA4/workspace/test/main.cpp:5-24 + test/test_spmm.cu:8-62 (executable `unit_tests`).
//
//   unit_tests --dataset D --datadir DIR --len N        (flags of src/util.cu:17-20)
//
// Same flow: argParse -> load_graph (course format incl. .ptrdump/.edgedump caches,
// src/data.cu:3-66) -> device CSR structure -> fixture allocates B, C, C_ref and A's values with
// N(0, 0.1) (include/data.h:24-37, seed 123) -> three tests:
//   SpMMTest.validation            SpMMOpt vs the reference arithmetic, valid() < M*N/10000 + 1
//   SpMMTest.cusparse_performance  vendor comparator (rocSPARSE), getAverageTimeWithWarmUp
//   SpMMTest.opt_performance       SpMMOpt, getAverageTimeWithWarmUp
// and the same log lines (`dset = "..." (std::string)`, `time = ... (double)`, gtest's
// RUN/OK/PASSED) so the reference's plot.py regexes (plot.py:13-14) parse our logs: the first
// `time =` of a block is the vendor library, the second is opt (test_spmm.cu:46-62).
//
// SpMMTest.validation compares SpMMOpt with SpMMRef on the device like the reference (test_spmm.cu:33-43).
// TEST INFRASTRUCTURE: as a second check SpMMRef's result must equal the CPU oracle bit for bit
// (oracle/liboracle.so, restating spmm_ref.cu:3-17) -- the product never links it.
#define MI_SPMM_WITH_COMPARATOR
#include "spmm_adapter.hpp"

#include <sys/stat.h>

#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern "C" void oracle_spmm_omp(const int32_t *ptr, const int32_t *idx, const float *val, const float *vin,
                                int64_t ldb, float *vout, int64_t ldc, int32_t num_v, int32_t feat,
                                int32_t row_begin, int32_t row_end);
extern "C" void oracle_set_threads(int n);

static int kNumV = -1, kNumE = -1, kLen = 0;
static std::string basedir, dataset;
static int *gptr = nullptr, *gidx = nullptr;
static uint64_t g_alloc_counter = 0;  // successive allocations draw successive subsequences (one seeded generator, main.cpp:19-20)

#define DBG_LINE(expr_str, fmt, val, type) \
    std::fprintf(stderr, "[tests/native/unit_tests.cpp:%d (%s)] %s = " fmt " (%s)\n", __LINE__, __func__, expr_str, val, type)

static bool fexist(const std::string &n)
{
    struct stat b;
    return stat(n.c_str(), &b) == 0;
}

// src/util.cu:14-76 (the three flags are required, as there)
static void argParse(int argc, char **argv)
{
    std::string dset, datadir;
    for (int i = 1; i < argc; ++i) {
        auto take = [&](const char *flag, std::string &dst) {
            const size_t L = std::strlen(flag);
            if (!std::strncmp(argv[i], flag, L)) {
                if (argv[i][L] == '=') { dst = argv[i] + L + 1; return true; }
                if (argv[i][L] == 0 && i + 1 < argc) { dst = argv[++i]; return true; }
            }
            return false;
        };
        std::string len;
        if (take("--dataset", dset) || take("--datadir", datadir)) continue;
        if (take("--len", len)) { kLen = std::atoi(len.c_str()); continue; }
        std::fprintf(stderr, "unknown flag %s\n", argv[i]);
        std::exit(1);
    }
    if (dset.empty() || datadir.empty() || kLen <= 0) {
        std::fprintf(stderr, "usage: unit_tests --dataset D --datadir DIR --len N\n");
        std::exit(1);
    }
    basedir = datadir;
    if (basedir.back() != '/') basedir += "/";
    dataset = dset;
    const std::string configpath = basedir + dset + ".config";
    if (!fexist(configpath)) { std::fprintf(stderr, "missing %s\n", configpath.c_str()); std::exit(1); }
    FILE *fin = std::fopen(configpath.c_str(), "r");
    if (std::fscanf(fin, "%d", &kNumV) != 1 || std::fscanf(fin, "%d", &kNumE) != 1) std::exit(1);
    std::fclose(fin);
    std::fprintf(stderr, "[tests/native/unit_tests.cpp:%d (argParse)] dset = \"%s\" (std::string)\n", __LINE__, dset.c_str());
}

// src/data.cu:3-66
static void load_graph(int num_v, int num_e, std::vector<int> &indptr, std::vector<int> &indices)
{
    std::fprintf(stderr, "[tests/native/unit_tests.cpp:%d (load_graph)] loading\n", __LINE__);
    const std::string inputgraph = basedir + dataset + ".graph";
    const std::string ptrfile = inputgraph + ".ptrdump", edgefile = inputgraph + ".edgedump";
    indptr.resize((size_t)num_v + 1);
    indices.resize((size_t)num_e);
    FILE *text = nullptr;
    bool text_at_edges = false;
    if (fexist(ptrfile)) {
        FILE *f = std::fopen(ptrfile.c_str(), "r");
        if (std::fread(indptr.data(), sizeof(int) * ((size_t)num_v + 1), 1, f) != 1) std::exit(1);
        std::fclose(f);
    } else {
        text = std::fopen(inputgraph.c_str(), "r");
        if (!text) { std::fprintf(stderr, "missing %s\n", inputgraph.c_str()); std::exit(1); }
        for (int i = 0; i < num_v + 1; ++i)
            if (std::fscanf(text, "%d", &indptr[i]) != 1) std::exit(1);
        text_at_edges = true;
        FILE *f = std::fopen(ptrfile.c_str(), "w");
        if (f) { std::fwrite(indptr.data(), sizeof(int) * ((size_t)num_v + 1), 1, f); std::fclose(f); }
    }
    if (indptr[num_v] != num_e) { std::fprintf(stderr, "indptr[num_v]=%d != num_e=%d\n", indptr[num_v], num_e); std::exit(1); }
    if (fexist(edgefile)) {
        FILE *f = std::fopen(edgefile.c_str(), "r");
        if (num_e && std::fread(indices.data(), sizeof(int) * (size_t)num_e, 1, f) != 1) std::exit(1);
        std::fclose(f);
    } else {
        if (!text) {  // only the ptr cache existed: re-open and skip the pointer section (the reference reads a closed FILE* here)
            text = std::fopen(inputgraph.c_str(), "r");
            if (!text) std::exit(1);
        }
        if (!text_at_edges) { int skip; for (int i = 0; i < num_v + 1; ++i) if (std::fscanf(text, "%d", &skip) != 1) std::exit(1); }
        for (int i = 0; i < num_e; ++i)
            if (std::fscanf(text, "%d", &indices[i]) != 1) std::exit(1);
        FILE *f = std::fopen(edgefile.c_str(), "w");
        if (f) { std::fwrite(indices.data(), sizeof(int) * (size_t)num_e, 1, f); std::fclose(f); }
    }
    if (text) std::fclose(text);
}

// include/data.h:24-37
static float *allocate(int num, std::vector<void *> *tensor_ptr, bool random = true)
{
    float *tmp = nullptr;
    const size_t n = ((size_t)num + 511) / 512 * 512;
    MI_CHECK(hipMalloc((void **)&tmp, sizeof(float) * n));
    if (random) MI_CHECK(mi_spmm_fill_normal(tmp, (int64_t)n, 123ULL, g_alloc_counter++, 0.f, 0.1f, nullptr));
    if (tensor_ptr) tensor_ptr->push_back(tmp);
    return tmp;
}

struct SpMMTest {  // test_spmm.cu:8-29
    std::vector<void *> tensor_ptr;
    float *p_in_feat_vec, *p_out_feat_vec, *p_out_feat_vec_ref, *p_value;
    CSR *g;
    void SetUp()
    {
        p_in_feat_vec = allocate(kNumV * kLen, &tensor_ptr);
        p_out_feat_vec = allocate(kNumV * kLen, &tensor_ptr);
        p_out_feat_vec_ref = allocate(kNumV * kLen, &tensor_ptr);
        p_value = allocate(kNumE, &tensor_ptr);
        g = new CSR(kNumV, kNumE, gptr, gidx, p_value);
    }
    void TearDown()
    {
        for (auto p : tensor_ptr) (void)hipFree(p);
        delete g;
    }
};

static int failures = 0;
#define RUN_TEST(name)                                                                          \
    do {                                                                                        \
        std::printf("[ RUN      ] SpMMTest." #name "\n");                                       \
        SpMMTest t;                                                                             \
        t.SetUp();                                                                              \
        auto t0 = std::chrono::steady_clock::now();                                             \
        const bool ok = name(t);                                                                \
        const long ms = (long)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); \
        t.TearDown();                                                                           \
        std::printf(ok ? "[       OK ] SpMMTest." #name " (%ld ms)\n" : "[  FAILED  ] SpMMTest." #name " (%ld ms)\n", ms); \
        if (!ok) ++failures;                                                                    \
    } while (0)

static std::vector<int> h_ptr, h_idx;

// test_spmm.cu:31-44 -- SpMMOpt against SpMMRef on the device, as the reference does; then both against the CPU
// restatement of spmm_kernel_ref (oracle) on the same device-generated inputs: SpMMRef must match it bit for bit.
static bool validation(SpMMTest &t)
{
    SpMM *spmmer_ref = new SpMMRef(t.g, kLen);
    spmmer_ref->preprocess(t.p_in_feat_vec, t.p_out_feat_vec_ref);
    SpMM *spmmer = new SpMMOpt(t.g, kLen);
    spmmer->preprocess(t.p_in_feat_vec, t.p_out_feat_vec);
    const size_t nB = (size_t)kNumV * kLen;
    MI_CHECK(hipMemset(t.p_out_feat_vec, 0, sizeof(float) * nB));
    MI_CHECK(hipMemset(t.p_out_feat_vec_ref, 0, sizeof(float) * nB));
    spmmer_ref->run(t.p_in_feat_vec, t.p_out_feat_vec_ref);
    spmmer->run(t.p_in_feat_vec, t.p_out_feat_vec);
    MI_CHECK(hipDeviceSynchronize());
    const int bad = valid(t.p_out_feat_vec, t.p_out_feat_vec_ref, kNumV * kLen);
    int64_t ndiff = -1;
    float maxabs = 0.f;
    MI_CHECK(mi_spmm_count_bitdiff(t.p_out_feat_vec, t.p_out_feat_vec_ref, (int64_t)nB, &ndiff, &maxabs, nullptr));
    std::fprintf(stderr, "[tests/native/unit_tests.cpp:%d (TestBody)] bad = %d (int)  bitdiff = %lld  maxabs = %g\n", __LINE__, bad,
                 (long long)ndiff, maxabs);
    // second check: the oracle (test infrastructure) on the same inputs -- SpMMRef is exact-order, so bit-identical
    std::vector<float> hB(nB), hVal((size_t)kNumE), hRef(nB), hGot(nB);
    MI_CHECK(hipMemcpy(hB.data(), t.p_in_feat_vec, nB * sizeof(float), hipMemcpyDeviceToHost));
    MI_CHECK(hipMemcpy(hVal.data(), t.p_value, (size_t)kNumE * sizeof(float), hipMemcpyDeviceToHost));
    MI_CHECK(hipMemcpy(hGot.data(), t.p_out_feat_vec_ref, nB * sizeof(float), hipMemcpyDeviceToHost));
    oracle_spmm_omp(h_ptr.data(), h_idx.data(), hVal.data(), hB.data(), kLen, hRef.data(), kLen, kNumV, kLen, 0, kNumV);
    const bool ref_exact = nB == 0 || std::memcmp(hGot.data(), hRef.data(), nB * sizeof(float)) == 0;
    std::fprintf(stderr, "[tests/native/unit_tests.cpp:%d (TestBody)] ref_vs_oracle_bit_identical = %d (int)\n", __LINE__, ref_exact ? 1 : 0);
    delete spmmer;
    delete spmmer_ref;
    return bad < kNumV * kLen / 10000 + 1 && ref_exact;  // ASSERT_LT, test_spmm.cu:43
}

// test_spmm.cu:46-53
static bool cusparse_performance(SpMMTest &t)
{
    SpMMCuSparse *spmmer = new SpMMCuSparse(t.g, kLen);
    spmmer->preprocess(t.p_in_feat_vec, t.p_out_feat_vec);
    const double time = getAverageTimeWithWarmUp([&]() { spmmer->run(t.p_in_feat_vec, t.p_out_feat_vec); });
    std::fprintf(stderr, "[tests/native/unit_tests.cpp:%d (TestBody)] time = %.9f (double)\n", __LINE__, time);
    delete spmmer;
    return true;
}

// test_spmm.cu:55-62
static bool opt_performance(SpMMTest &t)
{
    SpMMOpt *spmmer = new SpMMOpt(t.g, kLen);
    spmmer->preprocess(t.p_in_feat_vec, t.p_out_feat_vec);
    const double time = getAverageTimeWithWarmUp([&]() { spmmer->run(t.p_in_feat_vec, t.p_out_feat_vec); });
    std::fprintf(stderr, "[tests/native/unit_tests.cpp:%d (TestBody)] time = %.9f (double)\n", __LINE__, time);
    delete spmmer;
    return true;
}

int main(int argc, char **argv)
{
    argParse(argc, argv);
    load_graph(kNumV, kNumE, h_ptr, h_idx);
    MI_CHECK(hipMalloc((void **)&gptr, ((size_t)kNumV + 1) * sizeof(int)));
    MI_CHECK(hipMalloc((void **)&gidx, (size_t)(kNumE > 0 ? kNumE : 1) * sizeof(int)));
    MI_CHECK(hipMemcpy(gptr, h_ptr.data(), sizeof(int) * ((size_t)kNumV + 1), hipMemcpyHostToDevice));
    MI_CHECK(hipMemcpy(gidx, h_idx.data(), sizeof(int) * (size_t)kNumE, hipMemcpyHostToDevice));
    std::fprintf(stderr, "[tests/native/unit_tests.cpp:%d (main)] kLen = %d (int)\n", __LINE__, kLen);
    oracle_set_threads(16);
    std::printf("[==========] Running 3 tests from 1 test case.\n[----------] 3 tests from SpMMTest\n");
    RUN_TEST(validation);
    RUN_TEST(cusparse_performance);
    RUN_TEST(opt_performance);
    std::printf("[==========] 3 tests from 1 test case ran.\n");
    if (failures) std::printf("[  FAILED  ] %d tests.\n", failures);
    else std::printf("[  PASSED  ] 3 tests.\n");
    (void)hipFree(gptr);
    (void)hipFree(gidx);
    return failures ? 1 : 0;
}
