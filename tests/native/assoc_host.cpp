// TEST INFRASTRUCTURE ONLY: host build of classifier-pipeline_amd/csrc/cpx_assoc_core.h so
// that the association logic can be debugged against the oracle without a GPU
// (tests/test_assoc_host.py).  The product never loads this; it runs the HIP
// build of the same header (cpx_assoc.hip).
#include <vector>

#include "cpx_assoc_core.h"

extern "C" int assoc_host_clip(const cpx_track_params* p, int cap, int nproc, const int* ffc, const int* ncomp,
                               const cpx_component* comps, cpx_region* pool, cpx_track_record* tracks,
                               int* n_tracks, cpx_region* regions_out, int* region_counts) {
  using namespace cpx;
  std::vector<ActiveTrack> active(p->max_active_tracks);
  std::vector<RegionRec> regs(cap);
  std::vector<ScoreRec> scores((size_t)cap * p->max_active_tracks);
  std::vector<unsigned char> used(cap);
  AssocClip c;
  c.p = p;
  c.cap = cap;
  c.max_active = p->max_active_tracks;
  c.max_tracks = p->max_tracks;
  c.pool = pool;
  c.active = active.data();
  c.n_active = 0;
  c.tracks = tracks;
  c.n_tracks = 0;
  c.next_id = 1;
  c.regs = regs.data();
  c.scores = scores.data();
  c.used = used.data();
  c.status = 0;
  for (int t = 0; t < nproc; ++t) {
    int nreg = 0;
    if (ffc[t]) {
      c.n_active = 0;
    } else {
      nreg = build_regions(c, comps + (size_t)t * cap, ncomp[t], t, t > 0);
      for (int i = 0; i < nreg; ++i) regions_out[(size_t)t * cap + i] = c.regs[i];
      assoc_frame(c, nreg, t);
    }
    region_counts[t] = nreg;
  }
  *n_tracks = c.n_tracks;
  return c.status;
}

#include "cpx_final_core.h"

// host build of the end-of-clip code (trim / stats / rejects / plan) for one clip
extern "C" int final_host_clip(const cpx_filter_params* fp, const cpx_region* pool, const cpx_track_record* recs,
                               int n_tracks, const int* proc_ffc, const int* proc_idx, int max_frames,
                               cpx_track_summary* out, int* counts, cpx_region_ref* refs, int* track_offsets,
                               cpx_crop_req* reqs, int* sample_track, int* track_clip) {
  using namespace cpx;
  std::vector<double> d(2 * (size_t)max_frames + 16);
  std::vector<float> f((size_t)max_frames + 16);
  FinalScratch sc{d.data(), f.data()};
  finalize_clip(*fp, pool, recs, n_tracks, proc_ffc, 5, out, counts, sc);
  const int prefix[4] = {0, 0, 0, 0};
  std::vector<unsigned char> taken((size_t)max_frames + 16);
  plan_clip(*fp, pool, out, n_tracks, proc_ffc, proc_idx, 5, 0, prefix, refs, track_offsets, reqs, sample_track, track_clip,
            taken.data());
  track_offsets[counts[0]] = counts[1];
  return 0;
}

// np.median restatements of the end-of-clip statistics (cpx_final_core.h), for a direct check against NumPy
extern "C" double median_select_host(double* a, int n) { return cpx::np_median_select(a, n); }
