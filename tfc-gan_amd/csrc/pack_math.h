// Index math of the packed weight operand stream, shared by the GPU pack kernel and the host-side emulator.
#pragma once
#include "tfc_desc.h"

// unit index idx = ((gs*NB32 + nb)*64 + lane)  ->  output channel n, filter-tap mask, first input channel c0 of the 16-B unit
static inline __host__ __device__ void tfc_pack_locate(const TfcGather& d, int es, int NB32, int idx, int* n, int* mask, int* c0) {
  const int lane = idx & 63;
  const int rest = idx >> 6;
  const int nb = rest % NB32;
  int gs = rest / NB32;
  const int PB = tfc_pb(d.Cin_pad, es);
  const int UPP = PB >> 4;
  const int CK = PB / es;
  const int UE = 16 / es;
  int per_chunk = 0;
  for (int pl = 0; pl < d.nplanes; ++pl) per_chunk += tfc_nsub(d.plane[pl].ntaps, PB);
  const int cc = gs / per_chunk;
  gs -= cc * per_chunk;
  int pl = 0;
  while (gs >= tfc_nsub(d.plane[pl].ntaps, PB)) { gs -= tfc_nsub(d.plane[pl].ntaps, PB); ++pl; }
  const int rn = lane & 31, h = lane >> 5;
  const int u = 2 * gs + h;
  const int tap = u / UPP, g = u % UPP;
  *n = nb * 32 + rn;
  *mask = d.plane[pl].tap_mask[tap];
  *c0 = cc * CK + g * UE;
}
