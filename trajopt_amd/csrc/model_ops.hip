// One compiled model: every kernel and launcher instantiated for MPPI_MODEL (a struct of models_gen.hpp),
// exported as the table mppi_ops_<model>.  Built once per model, in parallel (see the Makefile).
#include "kernels.hpp"

#define MPPI_CAT2(a, b) a##b
#define MPPI_CAT(a, b) MPPI_CAT2(a, b)
extern const ModelOps MPPI_CAT(mppi_ops_, MPPI_MODEL);
const ModelOps MPPI_CAT(mppi_ops_, MPPI_MODEL) = make_ops<MPPI_MODEL>();
