// lt_prep.hpp -- scene preparation on the device (lt_prep.hip): what lt_hip_set_scene derives from the caller's node buffer --
// the structural checks, the leaf order table of the reference's walk, the backend's own binned-SAH hierarchy over the caller's
// leaves and its collapse into 4-wide groups -- made by kernels from the uploaded buffers.  lt_retree.hpp holds the same
// algorithms on the host: they serve scenes this path declines (flags != 0 below), the diagnostic entry points of the C ABI,
// and the tests, which hold the two builds against each other node for node.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lt_prep {

enum : uint32_t {
  kFlagStructure = 1,    // an index out of range, a child that is not behind its parent: the host's validation words the error
  kFlagDuplicate = 2,    // two leaves on one primitive
  kFlagNotNested = 4,    // a child's box outside its parent's, a NaN, a bound beyond 2^40
  kFlagNotProper = 8,    // a node nobody refers to, or two parents: not the pre-order tree the scene builder writes
  kFlagTooDeep = 16,     // deeper than the reference's 64-entry stack
  kFlagPrimitives = 32,  // a materialIndex out of range
  kFlagInternal = 64,    // an invariant of the device build did not hold (never seen; the host build takes over)
  kFlagNoRoom = 128,     // the tree cannot fit under the height limit
};

struct Out {
  uint32_t flags = 0;           // != 0: take the host path (which also words the error, if there is one)
  int bvh_height = 0;           // of the caller's tree
  int own_height = -1;          // of the tree in d_nodes2
  int wide_height = -1;         // of the 4-wide group tree, -1: none
  uint32_t n_own = 0, groups = 0;
  float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};   // the own tree's root box
  void* d_nodes2 = nullptr;     // from the allocator, the caller's to give back: the own tree, 32-byte nodes in pre-order
  void* d_rank8 = nullptr;      // 8 x n_prims positions of the reference's walk (lt_retree::reference_order)
  void* d_children = nullptr;   // 4 x groups binary nodes (lt_retree::collapse_wide's `children`), allocated for 4 x (leaves - 1)
  void* d_groupOf = nullptr;    // n_own
  float ms_check = 0, ms_build = 0, ms_wide = 0;   // host wall time of the three stages (they end in a synchronisation each)
  int levels = 0;
};

// Where device memory comes from and goes to (lt_capi.hip keeps the buffers of the scene before for the next one of the same shape).
struct Allocator {
  void* self;
  hipError_t (*get)(void* self, void** p, size_t bytes);
  void (*put)(void* self, void* p);
};

// d_nodes / d_prims: the caller's buffers, already on the device (d_prims may be null: see check_primitives).  ownSplits = false: the caller's splits under the own
// structures (LT_RETREE=0).  wantWide = false: no 4-wide groups.  Returns a HIP error of a runtime call, or hipSuccess with
// out.flags telling whether the results are to be used.  Nothing is left allocated when flags != 0 or on an error.
hipError_t run(const void* d_nodes, uint32_t n_nodes, const void* d_prims, uint32_t n_prims, uint32_t n_mats, int maxHeight, int slack,
               bool ownSplits, hipStream_t stream, const Allocator& al, Out& out);

void release(Out& out, const Allocator& al);

// materialIndex of every primitive in range?  (d_prims == nullptr in run(): the caller checks them with this, e.g. because its
// primitives were still on their way while the hierarchy was built from the nodes.)
hipError_t check_primitives(const void* d_prims, uint32_t n_prims, uint32_t n_mats, hipStream_t stream, uint32_t* d_word, bool& ok);

}  // namespace lt_prep
