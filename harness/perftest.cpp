// perftest.cpp -- BER Monte-Carlo of the min-sum decoder, the native counterpart of the reference's perftest
// binary (perftest/src/main.rs), written against the C ABI only (include/labrador_ldpc_hip.h) plus the HIP
// runtime for device buffers.
//
// The reference: every rayon worker loops ms_trial (:9-29: random bytes -> encode -> hard_to_llrs -> add
// Normal(0, sigma) noise -> decode_ms with 100 iterations -> count bit errors in the first k bits) and adds
// (1 << 32 | errors) to one atomic until trials * k > 5e7 or errors > 5000 (:39-52); one CSV line per SNR
// (:62):  code,snr,trials,bits,errors,ber   with errors floored at 1 (:59-61).
// Here a WORKER IS A GPU: one host thread per device (the spawn_broadcast of :39) loops over trial BATCHES --
// random bytes on the host, labrador_ldpc_encode_batch, labrador_ldpc_hip_awgn_f32 and
// labrador_ldpc_decode_ms_batch_f32 on device-resident buffers, error counting on the host -- and adds to the
// same two counters until the same stopping rule fires.
//
//   perftest [CODE] [--noise perftest|ebn0] [--snrs 0.8,0.9,...] [--maxiters 100] [--batch 65536]
//            [--max-bits 5e7] [--max-errors 5000] [--devices all|0,1,...] [--seed 1]
// Noise: perftest = the reference's sigma = 10^(-snr/10) (:15); ebn0 = the textbook
// sigma^2 = 1 / (2 R 10^(EbN0/10)).
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "labrador_ldpc.h"

namespace {

const char *const NAMES[9] = {"TC128", "TC256", "TC512", "TM1280", "TM1536", "TM2048", "TM5120", "TM6144", "TM8192"};

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { std::fprintf(stderr, "perftest: %s: %s\n", #expr, hipGetErrorString(e_)); std::exit(2); } } while (0)
#define LDPC_OK(expr) do { int s_ = (expr); if (s_ != 0) { std::fprintf(stderr, "perftest: %s: status %d: %s\n", #expr, s_, labrador_ldpc_hip_last_error()); std::exit(2); } } while (0)

struct Args {
    enum labrador_ldpc_code code = LABRADOR_LDPC_CODE_TC512;             // perftest/src/main.rs:69
    bool ebn0 = false;
    std::vector<float> snrs{0.8f, 0.9f, 1.0f, 1.1f, 1.2f, 1.3f, 1.4f, 1.5f, 1.6f, 1.7f, 1.8f, 1.9f, 2.0f, 2.1f, 2.2f};   // :68
    size_t maxiters = 100, batch = 65536;                                  // :22
    double max_bits = 5e7;                                                 // :50
    uint64_t max_errors = 5000, seed = 1;
    std::vector<int> devices;
};

std::vector<float> parse_floats(const char *s)
{
    std::vector<float> v;
    for (const char *p = s; *p;) { char *e; v.push_back(std::strtof(p, &e)); p = *e ? e + 1 : e; }
    return v;
}

// one GPU's share of one SNR point
void worker(const Args &a, int device, float sigma, uint64_t stream_id, std::atomic<uint64_t> &trials,
            std::atomic<uint64_t> &errors, std::atomic<bool> &term)
{
    HIP_OK(hipSetDevice(device));
    const size_t n = labrador_ldpc_code_n(a.code), k = labrador_ldpc_code_k(a.code), out_len = labrador_ldpc_output_len(a.code);
    const size_t B = a.batch;
    uint8_t *d_data, *d_cw, *d_out, *d_ok;
    uint32_t *d_it;
    float *d_llrs;
    HIP_OK(hipMalloc((void **)&d_data, B * k / 8));
    HIP_OK(hipMalloc((void **)&d_cw, B * n / 8));
    HIP_OK(hipMalloc((void **)&d_llrs, B * n * sizeof(float)));
    HIP_OK(hipMalloc((void **)&d_out, B * out_len));
    HIP_OK(hipMalloc((void **)&d_it, B * sizeof(uint32_t)));
    HIP_OK(hipMalloc((void **)&d_ok, B));
    std::vector<uint8_t> data(B * k / 8), out(B * out_len);
    std::mt19937_64 rng(a.seed * 0x9E3779B97F4A7C15ull + stream_id);
    struct labrador_ldpc_hip_opts o;
    std::memset(&o, 0, sizeof o);
    o.device = device;
    o.memory = LABRADOR_LDPC_HIP_MEM_DEVICE;
    for (uint64_t round = 0; !term.load(std::memory_order_relaxed); ++round) {
        for (size_t i = 0; i + 8 <= data.size(); i += 8) { const uint64_t r = rng(); std::memcpy(&data[i], &r, 8); }   // :10-11
        HIP_OK(hipMemcpy(d_data, data.data(), data.size(), hipMemcpyHostToDevice));
        LDPC_OK(labrador_ldpc_encode_batch(a.code, d_data, d_cw, B, &o));                                               // :12
        LDPC_OK(labrador_ldpc_hip_awgn_f32(a.code, d_cw, B, d_llrs, B, sigma, (stream_id << 24) + round, &o));          // :13-18
        LDPC_OK(labrador_ldpc_decode_ms_batch_f32(a.code, d_llrs, d_out, d_it, d_ok, B, a.maxiters, &o));               // :22
        HIP_OK(hipMemcpy(out.data(), d_out, out.size(), hipMemcpyDeviceToHost));
        uint64_t errs = 0;
        for (size_t f = 0; f < B; ++f)                                                                                   // :23-28
            for (size_t j = 0; j < k / 8; ++j) errs += (uint64_t)__builtin_popcount((unsigned)(data[f * (k / 8) + j] ^ out[f * out_len + j]));
        trials.fetch_add(B, std::memory_order_relaxed);
        errors.fetch_add(errs, std::memory_order_relaxed);
    }
    for (void *p : {(void *)d_data, (void *)d_cw, (void *)d_llrs, (void *)d_out, (void *)d_it, (void *)d_ok}) (void)hipFree(p);
}

}  // namespace

int main(int argc, char **argv)
{
    Args a;
    bool all_devices = false;
    for (int i = 1; i < argc; ++i) {
        const std::string s = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "perftest: %s needs a value\n", s.c_str()); std::exit(1); } return argv[++i]; };
        if (s == "--noise") a.ebn0 = std::string(next()) == "ebn0";
        else if (s == "--snrs") a.snrs = parse_floats(next());
        else if (s == "--maxiters") a.maxiters = (size_t)std::atol(next());
        else if (s == "--batch") a.batch = (size_t)std::atol(next());
        else if (s == "--max-bits") a.max_bits = std::atof(next());
        else if (s == "--max-errors") a.max_errors = (uint64_t)std::atof(next());
        else if (s == "--seed") a.seed = (uint64_t)std::atoll(next());
        else if (s == "--devices") { const std::string v = next(); if (v == "all") all_devices = true; else for (float d : parse_floats(v.c_str())) a.devices.push_back((int)d); }
        else {
            int c = -1;
            for (int j = 0; j < 9; ++j) if (s == NAMES[j]) c = j;
            if (c < 0) { std::fprintf(stderr, "perftest: unknown argument %s\n", s.c_str()); return 1; }
            a.code = (enum labrador_ldpc_code)c;
        }
    }
    const int ndev = labrador_ldpc_hip_device_count();
    if (ndev == 0) { std::fprintf(stderr, "perftest: no gfx950 device (the decoder has no CPU path)\n"); return 77; }
    if (all_devices) for (int d = 0; d < ndev; ++d) a.devices.push_back(d);
    if (a.devices.empty()) a.devices.push_back(0);
    a.batch &= ~(size_t)7;
    if (a.batch == 0) a.batch = 8;
    const size_t k = labrador_ldpc_code_k(a.code), n = labrador_ldpc_code_n(a.code);
    for (size_t si = 0; si < a.snrs.size(); ++si) {
        const float snr = a.snrs[si];
        const float sigma = a.ebn0 ? (float)std::sqrt(1.0 / (2.0 * ((double)k / n) * std::pow(10.0, snr / 10.0)))
                                   : (float)(1.0 / std::pow(10.0, snr / 10.0));                    // :15
        std::atomic<uint64_t> trials{0}, errors{0};
        std::atomic<bool> term{false};
        std::vector<std::thread> pool;
        for (size_t w = 0; w < a.devices.size(); ++w)                                               // :39
            pool.emplace_back(worker, std::cref(a), a.devices[w], sigma, (uint64_t)(si * 64 + w + 1), std::ref(trials), std::ref(errors), std::ref(term));
        for (;;) {                                                                                  // :44-52
            if ((double)trials.load() * (double)k > a.max_bits || errors.load() > a.max_errors) { term.store(true); break; }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        for (auto &t : pool) t.join();
        const uint64_t nt = trials.load(), ne = errors.load() > 0 ? errors.load() : 1;              // :57-59
        std::printf("%s,%.2f,%llu,%llu,%llu,%.5e\n", NAMES[a.code], snr, (unsigned long long)nt, (unsigned long long)(nt * k),
                    (unsigned long long)ne, (double)ne / ((double)k * (double)nt));                  // :62
        std::fflush(stdout);
    }
    return 0;
}
