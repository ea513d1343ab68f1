// csrc/search_inst.h -- the list of k_search / k_search_wave instantiations. Included twice: by search_kernels.hip with
// PCU_SEARCH_INST = `template` (explicit instantiation: the kernels are compiled there) and by pcu_hip.hip with
// PCU_SEARCH_INST = `extern template` (no second copy: the launches bind to the other unit's kernels at link time).
// The K values are the list capacities launch_search / launch_search_wave dispatch on (pcu_hip.hip).
#define PCU_LANE(T, K) PCU_SEARCH_INST __global__ void k_search<T, K>(const SearchArgs<T>);
#define PCU_RUNS(T, K) PCU_SEARCH_INST __global__ void k_search_runs<T, K>(const SearchArgs<T>);
#define PCU_WAVE(T, K) PCU_SEARCH_INST __global__ void k_search_wave<T, K>(const SearchArgs<T>, const SearchArgs<T>, int, const int);
PCU_LANE(float, 1) PCU_LANE(float, 4) PCU_LANE(float, 8) PCU_LANE(float, 16) PCU_LANE(float, 32)
PCU_LANE(double, 1) PCU_LANE(double, 4) PCU_LANE(double, 8) PCU_LANE(double, 16) PCU_LANE(double, 32)
PCU_RUNS(float, 4) PCU_RUNS(float, 8) PCU_RUNS(float, 16) PCU_RUNS(float, 32)
PCU_RUNS(double, 4) PCU_RUNS(double, 8) PCU_RUNS(double, 16) PCU_RUNS(double, 32)
PCU_WAVE(float, 2) PCU_WAVE(float, 4) PCU_WAVE(float, 8) PCU_WAVE(float, 16) PCU_WAVE(float, 32) PCU_WAVE(float, 64) PCU_WAVE(float, 128)
PCU_WAVE(double, 2) PCU_WAVE(double, 4) PCU_WAVE(double, 8) PCU_WAVE(double, 16) PCU_WAVE(double, 32) PCU_WAVE(double, 64) PCU_WAVE(double, 128)
#undef PCU_LANE
#undef PCU_RUNS
#undef PCU_WAVE
