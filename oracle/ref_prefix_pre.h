// oracle/ref_prefix_pre.h -- TEST INFRASTRUCTURE ONLY (oracle/_ref/libaasm_ref_prefix*.so).
//
// Own code.  Force-included (g++ -include) ahead of the stream that oracle/Makefile pipes into
// g++: the REAL /root/reference/src/paf_data.cpp from its first line up to (not including) the
// first line that names an ankerl type (`using IntBoolMap = ankerl...`, :739), minus the one
// `#include <ankerl/...>` (:13), followed by oracle/ref_prefix_epilogue.inc.  That prefix is
// solve_ctg_read() from the sort through k_shortest_walks() -- K1 ... K8 of SURVEY.md s8(a) --
// and uses nothing this image lacks.  Nothing is substituted and no stand-in header exists:
// this file only
//   (a) declares the store the epilogue copies the function's locals into, and
//   (b) includes the reference's own k_shortest_walks.hpp once with `private` lifted (same
//       device as oracle/ref_harness.cpp), so the epilogue can read the solver's d / best / h /
//       alloc / nodes / prev_node / path_last_node; the include guard makes paf_data.cpp's own
//       include of that header a no-op.
#pragma once
#include <algorithm>
#include <cassert>
#include <cinttypes>
#include <cstdint>
#include <deque>
#include <iostream>
#include <map>
#include <queue>
#include <string>
#include <string_view>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#include "paf_data.hpp"
#include "graph_operations.hpp"
#include "k_weighted_bfs.hpp"
#include "leftist_heap.hpp"
#define private public
#include "k_shortest_walks.hpp"
#undef private

namespace refp {
struct Dump {
    std::map<std::string, std::vector<int64_t>> arr;
    int64_t max_paths = 0;      // > 0: recover this many of the k walks (kth_shortest_walk_recover) into "path_*"
};
extern thread_local Dump *g_dump;   // null = no capture (timing runs)
}
