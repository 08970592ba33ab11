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
// The O(M*T) scan of the reference becomes a hash join on the coordinate bit patterns: observations are nodes of
// per-track singly linked lists in one pool, the live list is rebuilt per keyframe, the join table is open-addressed.
#include <cstdint>
#include <atomic>
#include <cstring>
#include <thread>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../include/meatmodeler.h"

namespace {

struct Node {
    int32_t frame, kp;
    int64_t prev;  // previous observation of the same track (-1 = first)
};

struct TrackRec {
    int64_t tail;  // last node
    int32_t len;
};

inline uint64_t coord_key(const float *xy) {
    float x = xy[0] + 0.0f, y = xy[1] + 0.0f;  // -0.0 -> +0.0 so that bit equality matches ==
    uint32_t a, b;
    memcpy(&a, &x, 4);
    memcpy(&b, &y, 4);
    return ((uint64_t)a << 32) | b;
}

inline uint64_t mix(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

}  // namespace

extern "C" {

int64_t mm_link_tracks_clip(int n_frames, int cap, const int32_t *kp_count, const float *kp_xy, int mcap,
                            const int32_t *match_count, const int32_t *matches, int64_t max_tracks, int64_t max_obs,
                            int64_t *track_ptr, int32_t *obs_frame, int32_t *obs_kp, int64_t *n_obs_out) {
    if (n_frames < 0 || cap < 0 || mcap < 0 || !n_obs_out) return MM_ERR_ARG;
    if (n_frames > 1 && (!kp_count || !kp_xy || !match_count || !matches)) return MM_ERR_ARG;
    std::vector<Node> nodes;
    std::vector<TrackRec> tracks;
    std::vector<int64_t> live, next, popped, fresh;
    std::vector<uint8_t> updated;
    std::vector<uint64_t> hkey;
    std::vector<int32_t> hval;  // position in `live`, -1 = empty
    int64_t total_matches = 0;
    for (int k = 0; k + 1 < n_frames; ++k) total_matches += match_count[k] > 0 ? match_count[k] : 0;
    nodes.reserve((size_t)total_matches * 2 + 16);
    tracks.reserve((size_t)total_matches + 16);
    for (int k = 0; k + 1 < n_frames; ++k) {
        const int M = match_count[k];
        if (M < 0 || M > mcap) return MM_ERR_ARG;
        // join table: coordinate at frame k -> FIRST position in `live`
        size_t tsz = 16;
        while (tsz < live.size() * 2 + 2) tsz <<= 1;
        hkey.assign(tsz, 0);
        hval.assign(tsz, -1);
        for (size_t pos = 0; pos < live.size(); ++pos) {
            const Node &tl = nodes[tracks[live[pos]].tail];  // every live track ends at frame k by construction
            if (tl.frame != k) continue;
            const uint64_t key = coord_key(kp_xy + ((size_t)k * cap + tl.kp) * 2);
            size_t h = mix(key) & (tsz - 1);
            bool present = false;
            while (hval[h] >= 0) {
                if (hkey[h] == key) {
                    present = true;  // an earlier (lower position) track already owns this coordinate
                    break;
                }
                h = (h + 1) & (tsz - 1);
            }
            if (!present) {
                hkey[h] = key;
                hval[h] = (int32_t)pos;
            }
        }
        updated.assign(live.size(), 0);
        fresh.clear();
        const int32_t *mk = matches + (size_t)k * mcap * 2;
        for (int m = 0; m < M; ++m) {
            const int q = mk[2 * m], tr = mk[2 * m + 1];
            if (q < 0 || q >= kp_count[k] || tr < 0 || tr >= kp_count[k + 1]) return MM_ERR_ARG;
            const uint64_t key = coord_key(kp_xy + ((size_t)k * cap + q) * 2);
            size_t h = mix(key) & (tsz - 1);
            int32_t pos = -1;
            while (hval[h] >= 0) {
                if (hkey[h] == key) {
                    pos = hval[h];
                    break;
                }
                h = (h + 1) & (tsz - 1);
            }
            if (pos >= 0) {
                TrackRec &t = tracks[live[pos]];
                if (nodes[t.tail].frame == k + 1) {
                    nodes[t.tail].kp = tr;  // second update in the same call overwrites the coordinate
                } else {
                    nodes.push_back(Node{k + 1, tr, t.tail});
                    t.tail = (int64_t)nodes.size() - 1;
                    t.len++;
                }
                updated[pos] = 1;
            } else {
                nodes.push_back(Node{k, q, -1});
                nodes.push_back(Node{k + 1, tr, (int64_t)nodes.size() - 1});
                tracks.push_back(TrackRec{(int64_t)nodes.size() - 1, 2});
                fresh.push_back((int64_t)tracks.size() - 1);
            }
        }
        next.clear();
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
    for (int64_t id : popped) n_obs += tracks[id].len;
    *n_obs_out = n_obs;
    const int64_t n_tracks = (int64_t)popped.size();
    if (n_tracks > max_tracks || n_obs > max_obs) return MM_ERR_WORKSPACE;
    if (n_tracks > 0 && (!track_ptr || !obs_frame || !obs_kp)) return MM_ERR_ARG;
    int64_t o = 0;
    for (int64_t i = 0; i < n_tracks; ++i) {
        const TrackRec &t = tracks[popped[i]];
        track_ptr[i] = o;
        int64_t w = o + t.len - 1;  // walk the list backwards, fill forwards
        for (int64_t nd = t.tail; nd >= 0; nd = nodes[nd].prev, --w) {
            obs_frame[w] = nodes[nd].frame;
            obs_kp[w] = nodes[nd].kp;
        }
        o += t.len;
    }
    if (track_ptr) track_ptr[n_tracks] = o;
    return n_tracks;
}

// Co-observation pairs for the banded Schur kernel: every ordered pair (o, o2) of observations of one point with
// camera(o2) <= camera(o), grouped by block segment s = camera(o) * (span + 1) + (camera(o) - camera(o2)).
// Inside a segment the pairs appear in a fixed canonical order (camera CSR order of o, then point CSR order of o2),
// which fixes the summation order of every entry of the reduced camera system.
// Call with pair_o == NULL to get the number of pairs; seg_ptr has F * (span + 1) + 1 entries.
int64_t mm_ba_build_pairs(int F, int P, int64_t O, const int32_t *fi, const int32_t *pi, const int32_t *pt_ptr,
                          const int32_t *pt_obs, const int32_t *cam_ptr, const int32_t *cam_obs, int span,
                          int64_t *seg_ptr, int32_t *pair_o, int32_t *pair_o2, int64_t max_pairs) {
    if (F < 0 || P < 0 || O < 0 || span < 0 || !seg_ptr) return MM_ERR_ARG;
    if (O > 0 && (!fi || !pi || !pt_ptr || !pt_obs || !cam_ptr || !cam_obs)) return MM_ERR_ARG;
    const int64_t nseg = (int64_t)F * (span + 1);
    for (int64_t s = 0; s <= nseg; ++s) seg_ptr[s] = 0;
    // Both passes are independent per camera i (its segments i*(span+1)+d are touched by no other camera), so they
    // run on a few host threads over contiguous camera ranges; the result does not depend on the thread count.
    const int nthreads = F >= 64 ? 8 : 1;
    std::atomic<int> bad{0};
    auto for_cameras = [&](auto &&body) {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t) {
            const int lo = (int)((int64_t)F * t / nthreads), hi = (int)((int64_t)F * (t + 1) / nthreads);
            pool.emplace_back([&, lo, hi] {
                for (int i = lo; i < hi; ++i) body(i);
            });
        }
        for (auto &th : pool) th.join();
    };
    for_cameras([&](int i) {
        for (int e = cam_ptr[i]; e < cam_ptr[i + 1]; ++e) {
            const int p = pi[cam_obs[e]];
            for (int e2 = pt_ptr[p]; e2 < pt_ptr[p + 1]; ++e2) {
                const int d = i - fi[pt_obs[e2]];
                if (d < 0) continue;
                if (d > span) {
                    bad = 1;  // span too small for this problem
                    continue;
                }
                ++seg_ptr[(int64_t)i * (span + 1) + d + 1];
            }
        }
    });
    if (bad) return MM_ERR_ARG;
    for (int64_t s = 0; s < nseg; ++s) seg_ptr[s + 1] += seg_ptr[s];
    const int64_t n = seg_ptr[nseg];
    if (!pair_o || !pair_o2) return n;
    if (n > max_pairs) return MM_ERR_WORKSPACE;
    std::vector<int64_t> cur(seg_ptr, seg_ptr + nseg);
    for_cameras([&](int i) {
        for (int e = cam_ptr[i]; e < cam_ptr[i + 1]; ++e) {
            const int o = cam_obs[e];
            const int p = pi[o];
            for (int e2 = pt_ptr[p]; e2 < pt_ptr[p + 1]; ++e2) {
                const int o2 = pt_obs[e2];
                const int d = i - fi[o2];
                if (d < 0) continue;
                const int64_t w = cur[(int64_t)i * (span + 1) + d]++;
                pair_o[w] = o;
                pair_o2[w] = o2;
            }
        }
    });
    return n;
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

// Greedy minimum-distance selection of cv2.goodFeaturesToTrack (reference processor.py:104): the candidates arrive
// sorted by strength (descending; ties by y, x); one is accepted iff no accepted corner lies closer than min_distance
// (buckets of min_distance pixels, 3 x 3 neighbourhood), up to max_corners (0 = no limit).  Sequential by nature and
// short (it stops after max_corners acceptances): host code, as in OpenCV.  -> number of corners written to out [cap,2].
int mm_gftt_select(const int32_t *pos /*[n] y * w + x, sorted*/, int64_t n, int w, int h, int max_corners, double min_distance,
                   float *out, int cap) {
    if (n < 0 || w < 1 || h < 1 || cap < 0 || (n > 0 && !pos) || (cap > 0 && !out)) return MM_ERR_ARG;
    int m = 0;
    if (min_distance >= 1.0) {
        const int cell = (int)lrint(min_distance);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        std::vector<int32_t> head((size_t)gw * gh, -1), nxt, ax, ay;
        const double md2 = min_distance * min_distance;
        for (int64_t i = 0; i < n; ++i) {
            const int x = pos[i] % w, y = pos[i] / w, cx = x / cell, cy = y / cell;
            bool good = true;
            for (int yy = cy > 0 ? cy - 1 : 0; yy <= (cy + 1 < gh ? cy + 1 : gh - 1) && good; ++yy)
                for (int xx = cx > 0 ? cx - 1 : 0; xx <= (cx + 1 < gw ? cx + 1 : gw - 1) && good; ++xx)
                    for (int32_t e = head[(size_t)yy * gw + xx]; e >= 0; e = nxt[e]) {
                        const double dx = x - ax[e], dy = y - ay[e];
                        if (dx * dx + dy * dy < md2) {
                            good = false;
                            break;
                        }
                    }
            if (!good) continue;
            ax.push_back(x);
            ay.push_back(y);
            nxt.push_back(head[(size_t)cy * gw + cx]);
            head[(size_t)cy * gw + cx] = m;
            if (m < cap) {
                out[2 * m] = (float)x;
                out[2 * m + 1] = (float)y;
            }
            ++m;
            if ((max_corners > 0 && m >= max_corners) || m >= cap) break;
        }
    } else {
        for (int64_t i = 0; i < n && m < cap; ++i) {
            out[2 * m] = (float)(pos[i] % w);
            out[2 * m + 1] = (float)(pos[i] / w);
            ++m;
            if (max_corners > 0 && m >= max_corners) break;
        }
    }
    return m;
}

// Binary little-endian PLY with double x, y, z vertices: what PyntCloud(pd.DataFrame(points, columns=x,y,z)).to_file(
// path + "Cloud.ply") writes for a float64 frame (reference processor.py:480-485).  -> 0 ok.
int mm_write_ply(const char *path, const double *xyz /*[n,3]*/, int64_t n) {
    if (!path || n < 0 || (n > 0 && !xyz)) return MM_ERR_ARG;
    FILE *f = fopen(path, "wb");
    if (!f) return MM_ERR_ARG;
    fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %lld\nproperty double x\nproperty double y\nproperty double z\nend_header\n",
            (long long)n);
    const size_t wrote = n ? fwrite(xyz, sizeof(double) * 3, (size_t)n, f) : 0;
    const int rc = fclose(f);
    return (wrote == (size_t)n && rc == 0) ? MM_OK : MM_ERR_ARG;
}

}  // extern "C"
