// Host-side bookkeeping of the C ABI: track linking over a clip, flattening, BA index build.
// (Pure host code; compiled into libmeatmodeler_hip.so next to the kernels.)
//
// mm_link_tracks_clip reproduces processor.pointTracking (reference processor.py:190-243) called once per
// consecutive keyframe pair, followed by `popped_tracks += tracks` (processor.py:418):
//   * a match continues the FIRST live track (list order) whose coordinate at the previous keyframe equals the
//     match's previous-frame point exactly (float equality on both coordinates, processor.py:220);
//   * a later match hitting the same track overwrites that track's new-frame coordinate (Track.update, track.py:17-19);
//   * unmatched feature points spawn new tracks, appended after the surviving old ones in match order
//     (processor.py:226-241); tracks not updated in a call are popped in list order (processor.py:233-238).
// The O(M*T) scan of the reference becomes a hash join on the coordinate bit patterns.
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>
#include "../../include/meatmodeler.h"

namespace {

struct Obs {
    int32_t frame, kp;
};

inline uint64_t coord_key(const float *xy) {
    float x = xy[0] + 0.0f, y = xy[1] + 0.0f;  // -0.0 -> +0.0 so that bit equality matches ==
    uint32_t a, b;
    memcpy(&a, &x, 4);
    memcpy(&b, &y, 4);
    return ((uint64_t)a << 32) | b;
}

}  // namespace

extern "C" {

int64_t mm_link_tracks_clip(int n_frames, int cap, const int32_t *kp_count, const float *kp_xy, int mcap,
                            const int32_t *match_count, const int32_t *matches, int64_t max_tracks, int64_t max_obs,
                            int64_t *track_ptr, int32_t *obs_frame, int32_t *obs_kp, int64_t *n_obs_out) {
    if (n_frames < 0 || cap < 0 || mcap < 0 || !n_obs_out) return MM_ERR_ARG;
    if (n_frames > 1 && (!kp_count || !kp_xy || !match_count || !matches)) return MM_ERR_ARG;
    std::vector<std::vector<Obs>> tracks;  // all tracks ever created
    std::vector<int64_t> live, popped;     // indices into `tracks`
    std::vector<uint8_t> updated;
    std::unordered_map<uint64_t, int64_t> first_live;  // coordinate at prev frame -> position in `live`
    for (int k = 0; k + 1 < n_frames; ++k) {
        const int M = match_count[k];
        if (M < 0 || M > mcap) return MM_ERR_ARG;
        first_live.clear();
        first_live.reserve(live.size() * 2 + 16);
        for (size_t pos = 0; pos < live.size(); ++pos) {
            const std::vector<Obs> &t = tracks[live[pos]];
            // every live track was created or updated at frame k by construction; its coordinate there:
            const Obs *at = nullptr;
            for (auto it = t.rbegin(); it != t.rend(); ++it)
                if (it->frame == k) {
                    at = &*it;
                    break;
                }
            if (!at) continue;
            uint64_t key = coord_key(kp_xy + ((size_t)k * cap + at->kp) * 2);
            first_live.emplace(key, (int64_t)pos);  // emplace keeps the FIRST position
        }
        updated.assign(live.size(), 0);
        std::vector<int64_t> fresh;
        const int32_t *mk = matches + (size_t)k * mcap * 2;
        for (int m = 0; m < M; ++m) {
            const int q = mk[2 * m], tr = mk[2 * m + 1];
            if (q < 0 || q >= kp_count[k] || tr < 0 || tr >= kp_count[k + 1]) return MM_ERR_ARG;
            uint64_t key = coord_key(kp_xy + ((size_t)k * cap + q) * 2);
            auto hit = first_live.find(key);
            if (hit != first_live.end()) {
                std::vector<Obs> &t = tracks[live[hit->second]];
                if (t.back().frame == k + 1)
                    t.back().kp = tr;  // second update in the same call overwrites
                else
                    t.push_back(Obs{k + 1, tr});
                updated[hit->second] = 1;
            } else {
                tracks.push_back(std::vector<Obs>{Obs{k, q}, Obs{k + 1, tr}});
                fresh.push_back((int64_t)tracks.size() - 1);
            }
        }
        std::vector<int64_t> next;
        next.reserve(live.size() + fresh.size());
        for (size_t pos = 0; pos < live.size(); ++pos) {
            if (updated[pos])
                next.push_back(live[pos]);
            else
                popped.push_back(live[pos]);
        }
        next.insert(next.end(), fresh.begin(), fresh.end());
        live.swap(next);
    }
    popped.insert(popped.end(), live.begin(), live.end());
    int64_t n_obs = 0;
    for (int64_t id : popped) n_obs += (int64_t)tracks[id].size();
    *n_obs_out = n_obs;
    const int64_t n_tracks = (int64_t)popped.size();
    if (n_tracks > max_tracks || n_obs > max_obs) return MM_ERR_WORKSPACE;
    if (n_tracks > 0 && (!track_ptr || !obs_frame || !obs_kp)) return MM_ERR_ARG;
    int64_t o = 0;
    for (int64_t i = 0; i < n_tracks; ++i) {
        track_ptr[i] = o;
        for (const Obs &ob : tracks[popped[i]]) {
            obs_frame[o] = ob.frame;
            obs_kp[o] = ob.kp;
            ++o;
        }
    }
    if (track_ptr) track_ptr[n_tracks] = o;
    return n_tracks;
}

int mm_ba_build_index(int F, int P, int64_t O, const int32_t *fi, const int32_t *pi, int32_t *pt_ptr, int32_t *pt_obs,
                      int32_t *cam_ptr, int32_t *cam_obs) {
    if (F < 0 || P < 0 || O < 0 || O > 0x7fffffff) return MM_ERR_ARG;
    if (O > 0 && (!fi || !pi)) return MM_ERR_ARG;
    if (!pt_ptr || !cam_ptr || (O > 0 && (!pt_obs || !cam_obs))) return MM_ERR_ARG;
    for (int i = 0; i <= P; ++i) pt_ptr[i] = 0;
    for (int i = 0; i <= F; ++i) cam_ptr[i] = 0;
    for (int64_t o = 0; o < O; ++o) {
        if (fi[o] < 0 || fi[o] >= F || pi[o] < 0 || pi[o] >= P) return MM_ERR_ARG;
        ++pt_ptr[pi[o] + 1];
        ++cam_ptr[fi[o] + 1];
    }
    for (int i = 0; i < P; ++i) pt_ptr[i + 1] += pt_ptr[i];
    for (int i = 0; i < F; ++i) cam_ptr[i + 1] += cam_ptr[i];
    std::vector<int32_t> pw(pt_ptr, pt_ptr + P), cw(cam_ptr, cam_ptr + F);
    for (int64_t o = 0; o < O; ++o) {  // stable: observation order is kept inside each segment
        pt_obs[pw[pi[o]]++] = (int32_t)o;
        cam_obs[cw[fi[o]]++] = (int32_t)o;
    }
    return MM_OK;
}

}  // extern "C"
