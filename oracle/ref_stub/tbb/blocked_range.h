// Serial stand-in for oneTBB (not installed in this image).  Only used to COMPILE the reference's
// own octree.h verbatim into oracle/_ref/: TBB is reached there only for octants holding >= 100000
// points (octree.h:561,626-651) and a serial execution is semantically identical.
#pragma once
namespace tbb {
template <class T>
struct blocked_range {
  T b, e;
  blocked_range(T b_, T e_) : b(b_), e(e_) {}
  T begin() const { return b; }
  T end() const { return e; }
};
}  // namespace tbb
