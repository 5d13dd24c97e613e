// Host-side helpers of the many-pairs path (no device code): the work list of a batch -- chunks, spans and per-pair record
// offsets (include/sp_hip.h, "Work list") -- from the padded run lengths of its segments.  The numpy form of the same
// (optim/batch_prepare.py flat_work_list) costs ~1-2 ms per lattice for 128 pairs x 64 segments, which made the HOST the bottleneck
// of the set-up once the preparation kernels were down to ~3 ms; these loops take tens of microseconds.
#include <stdint.h>
#include "../../include/sp_hip.h"

namespace {
inline int64_t chunk_max_of(int tile_points, int64_t granule) {
    const int64_t c = (int64_t)tile_points / granule * granule;
    return c > granule ? c : granule;
}
}  // namespace

extern "C" {

int sp_host_layout(const int32_t* counts, int n_lattices, int n_segs, const long long* n_off, int n_pairs, int granule,
                   long long* pc, long long* seg_pos, long long* p_off, int32_t* seg_off, long long* points) {
    if (!counts || !n_off || !pc || !seg_pos || !p_off || !seg_off || !points) return SP_EINVAL;
    if (n_lattices <= 0 || n_segs < 0 || n_pairs < 0 || (granule != 64 && granule != 256) || n_off[n_pairs] != n_segs) return SP_EINVAL;
    const int64_t g = granule, S = n_segs, M = n_pairs;
    for (int l = 0; l < n_lattices; ++l) {
        const int32_t* c = counts + (int64_t)l * S;
        long long* pcl = pc + (int64_t)l * S;
        long long* spl = seg_pos + (int64_t)l * S;
        long long* pol = p_off + (int64_t)l * (M + 1);
        int32_t* flat = seg_off + (int64_t)l * 2 * S;
        int32_t* rel = flat + S;
        int64_t at = 0;
        for (int64_t m = 0; m < M; ++m) {
            const int64_t first = at;
            int64_t real = 0;
            pol[m] = at;
            for (int64_t s = n_off[m]; s < n_off[m + 1]; ++s) {
                const int64_t p = ((int64_t)c[s] + g - 1) / g * g;
                pcl[s] = p;
                spl[s] = at - first;
                if (at > 0x7fffffffLL) return SP_ELIMIT;
                flat[s] = (int32_t)at;
                rel[s] = (int32_t)(at - first);
                real += c[s];
                at += p;
            }
            points[(int64_t)l * M + m] = real;
        }
        pol[M] = at;
    }
    return 0;
}

int sp_host_work_list_chunks(const long long* pc, int n_segs, int tile_points, int granule) {
    if (!pc || n_segs < 0 || tile_points <= 0 || (granule != 64 && granule != 256)) return SP_EINVAL;
    const int64_t cm = chunk_max_of(tile_points, granule);
    long long total = 0;
    for (int s = 0; s < n_segs; ++s) total += (pc[s] + cm - 1) / cm;
    return total > 0x7fffffffLL ? SP_ELIMIT : (int)total;
}

int sp_host_work_list(const long long* pc, const long long* seg_pos, const long long* n_off, int n_pairs, int span_points,
                      int tile_points, int granule, int records_per_chunk, int32_t* chunks, int32_t* spans, int32_t* seg_tile_off,
                      long long* sto_off, long long* c_off, long long* s_off) {
    if (!pc || !seg_pos || !n_off || !chunks || !spans || !seg_tile_off || !sto_off || !c_off || !s_off) return SP_EINVAL;
    if (n_pairs < 0 || span_points <= 0 || tile_points <= 0 || (granule != 64 && granule != 256) || records_per_chunk <= 0) return SP_EINVAL;
    const int64_t kGranule = granule, R = records_per_chunk;
    const int64_t cm = chunk_max_of(tile_points, granule);
    int64_t nc = 0, ns = 0;
    for (int m = 0; m < n_pairs; ++m) {
        const int64_t first = nc;
        c_off[m] = nc;
        s_off[m] = ns;
        sto_off[m] = n_off[m] + m;
        int32_t* sto = seg_tile_off + sto_off[m];
        for (int64_t s = n_off[m]; s < n_off[m + 1]; ++s) {
            *sto++ = (int32_t)(R * (nc - first));
            const int64_t p = pc[s];
            // pieces of (nearly) equal, granule-aligned length; a segment that fits one chunk -- the usual case -- without the divisions
            const int64_t k = p <= cm ? (p > 0) : (p + cm - 1) / cm;
            if (k > 0) {
                const int64_t per = k == 1 ? p : ((p / kGranule + k - 1) / k) * kGranule;
                for (int64_t done = 0; done < p; done += per) {
                    int32_t* c = chunks + 4 * nc++;
                    c[0] = m; c[1] = (int32_t)(s - n_off[m]); c[2] = (int32_t)(seg_pos[s] + done);
                    c[3] = (int32_t)(p - done < per ? p - done : per);
                }
            }
        }
        *sto = (int32_t)(R * (nc - first));
        // spans: greedy runs of consecutive chunks of this pair, at most span_points points each (at least one chunk)
        for (int64_t q = first; q < nc;) {
            int64_t q1 = q, pts = 0;
            while (q1 < nc && (q1 == q || pts + chunks[4 * q1 + 3] <= span_points)) pts += chunks[4 * q1++ + 3];
            int32_t* sp = spans + 4 * ns++;
            sp[0] = (int32_t)q; sp[1] = (int32_t)(q1 - q); sp[2] = (int32_t)pts; sp[3] = m;
            q = q1;
        }
    }
    c_off[n_pairs] = nc;
    s_off[n_pairs] = ns;
    sto_off[n_pairs] = n_off[n_pairs] + n_pairs;
    return (int)ns;
}

}  // extern "C"
