// Internal interface of the host-side numpy stream generator (numpy_stream.cpp) for the rest of the library.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>

// nstreams streams of n_each values (stream i: np.random.seed(seeds[i]), then normal draws) into out + i * n_each.
// ready (may be null): called with (stream, number of leading values of that stream that are final) whenever that
// number has grown, from the generator's threads (one call at a time per stream), so that a consumer can take the
// values away while the rest is being made.  state_out (may be null; one stream only): the
// 628-word generator state record described at mppi_numpy_normals.  Returns 0, -1 (bad argument) or -8; may throw
// std::bad_alloc.
int mppi_numpy_normals_streams(const uint32_t* seeds, int32_t nstreams, size_t n_each, double* out, int32_t nthreads,
                               uint32_t* state_out, const std::function<void(int, size_t)>* ready);
