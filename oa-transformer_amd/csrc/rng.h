// Counter-based dropout masks: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11;
// the generator behind torch.cuda's dropout).  A mask element is a pure function of
//   (seed, step offset, site, element index)
// so forward and backward regenerate the same mask and nothing is stored.  The reference's masks come from torch's
// own Philox stream (HF DistilBERT nn.Dropout, called from /root/reference/OATrans/model/oa_model.py:113-121); they
// cannot be reproduced bit for bit, the DISTRIBUTION (Bernoulli(1-p), scaled by 1/(1-p)) is what is matched.
#pragma once
#include "common.h"

namespace oat {

struct u32x4 { uint32_t x, y, z, w; };

OAT_DEV u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t lo0 = 0xD2511F53u * c.x, hi0 = __umulhi(0xD2511F53u, c.x);
    const uint32_t lo1 = 0xCD9E8D57u * c.z, hi1 = __umulhi(0xCD9E8D57u, c.z);
    c = u32x4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// rng[0] = seed, rng[1] = offset (advanced once per forward call by oat_rng_tick); both live in device memory so a
// captured hipGraph draws new masks on every replay.
struct DropSite {
  const unsigned long long* rng;
  uint32_t site;
  uint32_t thresh;      // drop when the 32-bit draw < thresh = p * 2^32
  float keep_scale;     // 1 / (1 - p)
};

// the 4 draws that cover elements [4q, 4q+4) of a site
OAT_DEV u32x4 drop_draw4(const DropSite& d, unsigned long long q) {
  const unsigned long long seed = d.rng[0], off = d.rng[1];
  return philox4x32_10(u32x4{(uint32_t)q, (uint32_t)(q >> 32), d.site, (uint32_t)off}, (uint32_t)seed, (uint32_t)(seed >> 32));
}
OAT_DEV float drop_mult(const DropSite& d, unsigned long long idx) {
  const u32x4 r = drop_draw4(d, idx >> 2);
  const uint32_t k = (uint32_t)idx & 3u;
  const uint32_t v = k == 0 ? r.x : k == 1 ? r.y : k == 2 ? r.z : r.w;
  return v < d.thresh ? 0.f : d.keep_scale;
}

inline DropSite make_drop_site(const void* rng, unsigned site, float p) {
  double t = (double)p * 4294967296.0;
  if (t > 4294967295.0) t = 4294967295.0;
  return DropSite{(const unsigned long long*)rng, site, (uint32_t)t, 1.0f / (1.0f - p)};
}

}  // namespace oat
