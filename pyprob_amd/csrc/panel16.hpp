// 16-row panel kernel for single-statement batches (panel16.hip): fragment images of the step's weight matrices and the
// host entry points used by engine.hip. Geometry, reasons and measurements: panel16.hip.
#pragma once
#include "common.hpp"
#include "panel.hpp"
#include "panel16_images.hpp"

namespace pp {

struct Panel16Args {
    PanelArgs a;                 // the 8-row kernel's arguments (WihT / W1T unused; xz / xd sized for FOUR workgroups per panel)
    const float* img[6];
};

// single-statement batch, one address, one LSTM layer, mixture head of <= 30 outputs, H = 512 | 1024, e = 64: the 16-row kernel
bool panel16_supported(int kind, int H, int hid, int n_out, int e, int B);
bool panel16_obs_ok(const ObsFusedArgs& oa);      // the embedding shapes its (MFMA) tail takes
int panel16(int kind, const Panel16Args& a, hipStream_t st, const PanelObs* obs = nullptr);
// granules (8 bytes) of the two hand-off buffers for B rows
int64_t panel16_xz_granules(int B, int H);
int64_t panel16_xd_granules(int B);

}  // namespace pp
