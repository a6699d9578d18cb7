// Operator boundary of the solver framework: the two abstract interfaces the
// reference declares in AbstractNonlinearProblem.hpp:6-14 and
// AbstractNonlinearProblemJacobian.hpp:6-13, with the same class names and
// virtual signatures so that a NewtonSolver written against the reference
// headers drives EventDrivenMap (event_driven_map.hpp) unchanged.
#pragma once
#include "mi355_arma_compat.hpp"

// Inside the reference's own tree (-DMI355_REFERENCE_TREE) the reference's two interface
// headers are used as they are, so that its NewtonSolver.o and this EventDrivenMap agree on the vtable; the include
// guards below are the reference's, which makes either order of inclusion work.
#if defined(MI355_REFERENCE_TREE)
#include "AbstractNonlinearProblem.hpp"
#include "AbstractNonlinearProblemJacobian.hpp"
#endif

#ifndef ABSTRACTCNONLINEARPROBLEMHEADERDEF
#define ABSTRACTCNONLINEARPROBLEMHEADERDEF
// residual F(u) of a nonlinear problem F(u) = 0
class AbstractNonlinearProblem {
  public:
    virtual ~AbstractNonlinearProblem() {}
    virtual void ComputeF(const arma::vec& u, arma::vec& f) = 0;
    // hook run by the solver once a Solve() has finished (default: nothing)
    virtual void PostProcess() {}
};
#endif

#ifndef ABSTRACTCNONLINEARPROBLEMJACOBIANHEADERDEF
#define ABSTRACTCNONLINEARPROBLEMJACOBIANHEADERDEF
// optional analytic Jacobian dF/du
class AbstractNonlinearProblemJacobian {
  public:
    virtual ~AbstractNonlinearProblemJacobian() {}
    virtual void ComputeDFDU(const arma::vec& u, arma::mat& dfdu) = 0;
};
#endif

// optional (not in the reference): several independent residual evaluations at once -- the columns of a
// finite-difference Jacobian.  Column j of F is what ComputeF(column j of U) returns.  A problem that implements it
// may overlap the evaluations on the device; NewtonSolver uses it when the problem offers it.
class AbstractBatchedNonlinearProblem {
  public:
    virtual ~AbstractBatchedNonlinearProblem() {}
    virtual void ComputeFBatch(const arma::mat& U, arma::mat& F) = 0;
};
