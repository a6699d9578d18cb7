#include "event_driven_map.hpp"

#include <cassert>
#include <chrono>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

namespace {

// the reference's CUDA_CALL convention: message to stderr, then exit(-1) (EventDrivenMap.cu:18-54)
void must(mi_status st, const mi_ctx* ctx, const char* what)
{
    if (st == MI_OK) return;
    std::fprintf(stderr, "%s failed: %s\n", what, mi_last_error(ctx));
    std::exit(-1);
}

void save_column(const std::string& path, size_t n, const float* x, const float* y = nullptr)
{
    FILE* fp = std::fopen(path.c_str(), "w");
    if (!fp) return;
    for (size_t i = 0; i < n; ++i) {
        if (y) std::fprintf(fp, "%f\t%f\n", x[i], y[i]);
        else std::fprintf(fp, "%f\n", x[i]);
    }
    std::fclose(fp);
}

template <typename T>
std::vector<float> as_float(const std::vector<T>& v)
{
    return std::vector<float>(v.begin(), v.end());
}

}  // namespace

EventDrivenMap::EventDrivenMap(const arma::vec* pParameters, unsigned int noReal, int device) : ctx_(nullptr), edm_(nullptr), device_(device)
{
    assert(pParameters && pParameters->n_elem >= 1);
    must(mi_ctx_create(device, &ctx_), nullptr, "mi_ctx_create");
    mi_edm_default_params(&p_);                         // parameters.hpp:1-15, N = 1024 (EventDrivenMap.cu:70)
    p_.beta_mean = static_cast<float>((*pParameters)[0]);   // conv_to<fvec>, EventDrivenMap.cu:61-62
    p_.n_real = noReal;
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    p_.seed = static_cast<unsigned long long>(now);      // mSeed = clock(), EventDrivenMap.cu:104
    must(mi_edm_create(ctx_, &p_, &edm_), ctx_, "mi_edm_create");
    partial_.set_size(MI_EDM_PARTIAL_LEN(p_.n_spikes));
    partial_.zeros();
}

EventDrivenMap::EventDrivenMap(const arma::vec* pParameters, unsigned int noReal, const std::vector<int>& devices)
    : ctx_(nullptr), edm_(nullptr), device_(devices.empty() ? 0 : devices[0])
{
    assert(pParameters && pParameters->n_elem >= 1 && !devices.empty());
    must(mi_group_create(static_cast<int>(devices.size()), devices.data(), &group_), nullptr, "mi_group_create");
    mi_edm_default_params(&p_);
    p_.beta_mean = static_cast<float>((*pParameters)[0]);
    p_.n_real = noReal;
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    p_.seed = static_cast<unsigned long long>(now);
    must(mi_group_edm_create(group_, &p_, &gedm_), nullptr, "mi_group_edm_create");
    ctx_ = mi_group_ctx(group_, 0);
    edm_ = mi_group_edm_shard(gedm_, 0);
    partial_.set_size(MI_EDM_PARTIAL_LEN(p_.n_spikes));
    partial_.zeros();
}

EventDrivenMap::~EventDrivenMap()
{
    for (Replica& r : replicas_) {
        mi_edm_destroy(r.edm);
        mi_ctx_destroy(r.ctx);
    }
    if (group_) {                         // the group owns the shard handles and contexts
        mi_group_edm_destroy(gedm_);
        mi_group_destroy(group_);
        return;
    }
    mi_edm_destroy(edm_);
    mi_ctx_destroy(ctx_);
}

void EventDrivenMap::Push()
{
    if (gedm_) must(mi_group_edm_set_params(gedm_, &p_), nullptr, "mi_group_edm_set_params");
    else must(mi_edm_set_params(edm_, &p_), ctx_, "mi_edm_set_params");
}

void EventDrivenMap::ComputeF(const arma::vec& u, arma::vec& f)
{
    assert(u.n_elem == p_.n_spikes);
    f.set_size(p_.n_spikes);
    partial_.set_size(MI_EDM_PARTIAL_LEN(p_.n_spikes));
    if (gedm_) must(mi_group_edm_compute_f(gedm_, u.memptr(), f.memptr(), partial_.memptr()), nullptr, "mi_group_edm_compute_f");
    else must(mi_edm_compute_f(edm_, u.memptr(), f.memptr(), partial_.memptr()), ctx_, "mi_edm_compute_f");
    if (debug_ && !gedm_) Dump();         // the Save* taps read one device's buffers: single-GPU mode only
}

void EventDrivenMap::ComputeFBatch(const arma::mat& U, arma::mat& F)
{
    const arma::uword S = p_.n_spikes, B = U.n_cols;
    assert(U.n_rows == S);
    F.set_size(S, B);
    if (debug_ || gedm_) {                              // the Save* dumps follow every single evaluation: keep them; a
                                                        // sharded problem already fills every GPU with one evaluation
        arma::vec u(S), f;
        for (arma::uword j = 0; j < B; ++j) {
            for (arma::uword i = 0; i < S; ++i) u(i) = U(i, j);
            ComputeF(u, f);
            for (arma::uword i = 0; i < S; ++i) F(i, j) = f(i);
        }
        return;
    }
    while (replicas_.size() < B) {                      // one context with a stream of its own per evaluation in flight
        Replica r;
        r.p = p_;
        must(mi_ctx_create(device_, &r.ctx), nullptr, "mi_ctx_create");
        must(mi_ctx_own_stream(r.ctx), r.ctx, "mi_ctx_own_stream");
        must(mi_edm_create(r.ctx, &r.p, &r.edm), r.ctx, "mi_edm_create");
        replicas_.push_back(r);
    }
    std::vector<double> z(S), f(S);
    for (arma::uword j = 0; j < B; ++j) {
        Replica& r = replicas_[j];
        if (std::memcmp(&r.p, &p_, sizeof(p_)) != 0) {   // a setter ran since the last batch
            r.p = p_;
            must(mi_edm_set_params(r.edm, &r.p), r.ctx, "mi_edm_set_params");
        }
        for (arma::uword i = 0; i < S; ++i) z[i] = U(i, j);
        must(mi_edm_compute_f_begin(r.edm, z.data()), r.ctx, "mi_edm_compute_f_begin");
    }
    for (arma::uword j = 0; j < B; ++j) {
        Replica& r = replicas_[j];
        must(mi_edm_compute_f_end(r.edm, f.data(), nullptr), r.ctx, "mi_edm_compute_f_end");
        for (arma::uword i = 0; i < S; ++i) F(i, j) = f[i];
    }
}

void EventDrivenMap::ResidualFromSums(const arma::vec& u, const arma::vec& sums_and_count, arma::vec& f) const
{
    f.set_size(p_.n_spikes);
    must(mi_edm_residual_from_sums(&p_, u.memptr(), sums_and_count.memptr(), f.memptr()), nullptr,
         "mi_edm_residual_from_sums");
}

void EventDrivenMap::SetTimeHorizon(const float T)
{
    assert(T > 0);
    p_.time_horizon = T;
    Push();
    if (!quiet_) std::cout << "Time horizon set to " << T << std::endl;
}

void EventDrivenMap::SetNoRealisations(const int noReal)
{
    assert(noReal > 0);
    p_.n_real = static_cast<unsigned int>(noReal);
    Push();
    if (!quiet_) std::cout << "Number of realisations set to " << noReal << std::endl;
}

void EventDrivenMap::SetNoThreads(const int noThreads)
{
    assert(noThreads > 0);
    assert(noThreads <= 1024);
    p_.n_grid = static_cast<unsigned int>(noThreads);
    Push();
    if (!quiet_) std::cout << "Number of threads set to " << noThreads << std::endl;
}

void EventDrivenMap::SetParameterStdDev(const float sigma)
{
    assert(sigma >= 0);
    p_.beta_stddev = sigma;
    Push();
    if (!quiet_) std::cout << "Parameter standard deviation set to " << sigma << std::endl;
}

void EventDrivenMap::SetParameters(const unsigned int parId, const float parVal)
{
    assert(parId == 0);               // the model has one parameter (beta), Driver.cu:15-16
    (void)parId;
    p_.beta_mean = parVal;
    Push();
    if (!quiet_) std::cout << "Parameter value set to " << parVal << std::endl;
}

void EventDrivenMap::ResetSeed() {}

void EventDrivenMap::SetSeed(unsigned long long seed)
{
    p_.seed = seed;
    Push();
}

void EventDrivenMap::SetNewSeed()
{
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    SetSeed(static_cast<unsigned long long>(now));
    if (!quiet_) std::cout << "New seed set" << std::endl;
}

void EventDrivenMap::PostProcess() { SetNewSeed(); }

void EventDrivenMap::SetDebugFlag(const bool val)
{
    debug_ = val;
    if (!quiet_) std::cout << (debug_ ? "Debugging on" : "Debugging off") << std::endl;
}

void EventDrivenMap::SetMathMode(int mode)
{
    p_.math_mode = mode;
    Push();
}

void EventDrivenMap::SetRealisationOffset(unsigned int offset)
{
    p_.real_offset = offset;
    Push();
}

void EventDrivenMap::SetMeanQuirk(bool on)
{
    p_.mean_quirk = on ? 1 : 0;
    Push();
    if (!quiet_) std::cout << (on ? "Averaging as the reference does (realisation 0 left out of the sum)" : "Averaging with the true mean") << std::endl;
}

void EventDrivenMap::SetDedupIdentical(bool on)
{
    p_.dedup_identical = on ? 1u : 0u;
    Push();
}

void EventDrivenMap::LastTimingsMs(float ms[4]) const { must(mi_edm_last_timings(edm_, ms), ctx_, "mi_edm_last_timings"); }

// The reference's only verification mechanism: one "%f" per line per stage (EventDrivenMap.cu:406-503, :911-917).
void EventDrivenMap::Dump()
{
    const size_t N = p_.n_grid, S = p_.n_spikes, R = p_.n_real, SR = S * R;
    std::vector<float> v(N), s(N), w(N), t0(SR), t1(SR), xr(SR);
    std::vector<uint16_t> i0(SR), i1(SR), seed(S);
    std::vector<uint32_t> acc(R);
    must(mi_edm_debug_read(edm_, v.data(), s.data(), w.data(), t0.data(), i0.data(), t1.data(), i1.data(), acc.data(),
                           xr.data(), seed.data()),
         ctx_, "mi_edm_debug_read");
    const std::string d = debug_dir_ + "/";
    save_column(d + "test.dat", N, w.data());                                   // coupling table, :122-127
    std::vector<float> seeds(SR);
    for (size_t m = 0; m < S; ++m)
        for (size_t r = 0; r < R; ++r) seeds[m * R + r] = static_cast<float>(seed[m]);
    save_column(d + "testInitLastSpikeInd.dat", SR, seeds.data());              // :406-420
    save_column(d + "testLift.dat", N, v.data(), s.data());                     // :422-436 (one profile, not R copies)
    save_column(d + "testLastSpikeTime.dat", SR, t0.data());                    // :438-483
    save_column(d + "testLastSpikeInd.dat", SR, as_float(i0).data());
    save_column(d + "testCrossedSpikeTime.dat", SR, t1.data());
    save_column(d + "testCrossedSpikeInd.dat", SR, as_float(i1).data());
    save_column(d + "testAcceptFlag.dat", R, as_float(acc).data());
    save_column(d + "testAverages.dat", SR, xr.data());                         // :485-493: Restrict output
    std::vector<float> mean(S);
    for (size_t m = 0; m < S; ++m) {   // the averaging rule of mi_edm_residual_from_sums
        const double sum = partial_[m] + ((p_.mean_quirk != 0 && partial_[S] == 1.0) ? partial_[S + 1 + m] : 0.0);
        mean[m] = static_cast<float>(sum / partial_[S]);
    }
    save_column(d + "testAveraged.dat", S, mean.data());                        // :495-503
}
