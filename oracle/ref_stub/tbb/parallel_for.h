// Serial stand-in for oneTBB -- see blocked_range.h in this directory.
#pragma once
namespace tbb {
template <class R, class F>
void parallel_for(const R& r, const F& f) { f(r); }
}  // namespace tbb
