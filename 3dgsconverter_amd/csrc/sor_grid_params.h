// sor_grid_params.h -- device-resident description of the KNN cell grid + work counters.
#pragma once

namespace gsx {

struct GridParams {
    float ox, oy, oz, inv_h;
    float h;
    int nx, ny, nz;
    int ncells;
    int nbx, nby, nbz;
    int nbricks;
    float tau1;      // f32 filter bound for r1sq
    double r1sq;     // (h' * (1 - 1e-3))^2, h' = 1/inv_h
    double hprime;   // 1/inv_h
    // work counters, zeroed by grid_params_kernel every call
    unsigned brick_next;
    unsigned fail_count;
    unsigned ring_next;
    unsigned exhaustive_count;
};

}  // namespace gsx
