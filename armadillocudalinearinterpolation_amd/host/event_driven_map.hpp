// EventDrivenMap: the reference's concrete problem (EventDrivenMap.hpp:11-121)
// as a thin C++ host class over the C ABI.  Same public interface, so
// `new EventDrivenMap(p_parameters, noReal)` and `NewtonSolver(p_event, ...)`
// read exactly like Driver.cu:20,34.  All device work (lift, evolve, restrict,
// average) happens in libmi355interp.so; this class only owns the handle, the
// parameter block and the optional debug dumps.
#pragma once
#include <string>
#include <vector>

#include "mi355_interp.h"
#include "nonlinear_problem.hpp"

class EventDrivenMap : public AbstractNonlinearProblem, public AbstractBatchedNonlinearProblem {
  public:
    EventDrivenMap(const arma::vec* pParameters, unsigned int noReal, int device = 0);
    // the same problem with the realisations sharded over several GPUs of the node (mi_group_edm_*): still ONE
    // ComputeF per residual (AbstractNonlinearProblem.hpp:11), so NewtonSolver and Driver stay as they are.  A repeated
    // ordinal (e.g. {0, 0}) rehearses the sharding on a box with fewer GPUs.
    EventDrivenMap(const arma::vec* pParameters, unsigned int noReal, const std::vector<int>& devices);
    virtual ~EventDrivenMap();
    EventDrivenMap(const EventDrivenMap&) = delete;              // the reference's copy would double-free
    EventDrivenMap& operator=(const EventDrivenMap&) = delete;

    // residual f(Z), Z = (c, Z1, .., Z_{S-1})   (EventDrivenMap.cu:154-240)
    void ComputeF(const arma::vec& u, arma::vec& f) override;
    // independent evaluations overlapped on the device: one replica (context + stream + buffers) per column
    void ComputeFBatch(const arma::mat& U, arma::mat& F) override;

    // setters of EventDrivenMap.hpp:27-51 (each echoes to stdout like the reference)
    void SetTimeHorizon(const float T);
    void SetNoRealisations(const int noReal);
    void SetNoThreads(const int noThreads);          // grid points; the reference asserts < 1024, here <= 1024
    void SetParameterStdDev(const float sigma);
    void SetParameters(const unsigned int parId, const float parVal);
    void ResetSeed();                                // no-op: the counter-based draw restarts every ComputeF
    void SetNewSeed();                               // clock-derived, like EventDrivenMap.cu:337-341
    void SetSeed(unsigned long long seed);           // reproducible alternative
    void PostProcess() override;                     // = SetNewSeed (EventDrivenMap.cu:343-346)
    void SetDebugFlag(const bool val);               // Save* dumps, EventDrivenMap.cu:406-503

    // Averaging over realisations.  true (default): exactly what the reference computes -- CountRealisationsKernel
    // overwrites accept[0] with the count (EventDrivenMap.cu:800-802), so the reduction (:817) leaves realisation 0 out
    // of the sum while :822 divides by the full count (unless only one realisation was accepted).  false: the true mean
    // over the accepted realisations.
    void SetMeanQuirk(bool on);
    bool MeanQuirk() const { return p_.mean_quirk != 0; }

    // extensions (not in the reference)
    void SetMathMode(int mode);                      // MI_EDM_MATH_EXACT / MI_EDM_MATH_FAST
    void SetRealisationOffset(unsigned int offset);  // this rank's first global realisation (multi-GPU shards)
    void SetDedupIdentical(bool on);                 // sigma == 0: evolve one realisation, replicate (bit-identical)
    void SetDebugDirectory(const std::string& dir) { debug_dir_ = dir; }
    void SetQuiet(bool q) { quiet_ = q; }
    // partial block of the last ComputeF, MI_EDM_PARTIAL_LEN(S) = 2S+1 values [un-normalised accepted sums | accepted
    // count | restricted position of realisation 0 (reference averaging, shard with offset 0)]: what an all-reduce sums
    const arma::vec& LastPartialSums() const { return partial_; }
    void ResidualFromSums(const arma::vec& u, const arma::vec& sums_and_count, arma::vec& f) const;
    const mi_edm_params& Parameters() const { return p_; }
    void LastTimingsMs(float ms[4]) const;

  private:
    void Push();
    void Dump();
    mi_ctx* ctx_;
    mi_edm* edm_;
    mi_group* group_ = nullptr;          // multi-GPU mode: the shards live in gedm_, edm_ / ctx_ alias shard 0 (debug taps)
    mi_group_edm* gedm_ = nullptr;
    mi_edm_params p_;
    struct Replica { mi_ctx* ctx; mi_edm* edm; mi_edm_params p; };
    std::vector<Replica> replicas_;
    int device_ = 0;
    arma::vec partial_;
    bool debug_ = false, quiet_ = false;
    std::string debug_dir_ = ".";
};
