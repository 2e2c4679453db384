// patch_match.h -- the reference's PatchMatch class (include/patch_match.h:41-103), same
// constructor and entry point, body replaced by the MI355X path behind the C ABI (pagk.h).
#pragma once
#include <vector>

#include "cvlite.h"
#include "pagk.h"

class GyroAidedTracker;

class PatchMatch {
public:
    // reference include/patch_match.h:44-49
    PatchMatch(GyroAidedTracker *pMatcher_, int halfPatchSize_, int iterations_, int pyramids_,
               bool bHasGyroPredictInitial_, bool bInverse_, bool bConsiderIllumination_,
               bool bConsiderAffineDeformation_, bool bRegularizationPenalty_ = true, bool bCalculateNCC_ = false);

    // Multi level optical flow tracking (reference src/patch_match.cpp:79-142): gathers the
    // tracker's inputs, runs pagk_track, scatters into the six result vectors (SetMatcher, :370-388).
    // Throws std::runtime_error when the HIP path cannot run: there is no CPU fallback.
    void OpticalFlowMultiLevel();

    // The context (device buffers, stream) this thread's PatchMatch instances share.
    static pagk_ctx *Context(int device = 0);
    static void ReleaseContext();

private:
    GyroAidedTracker *mpMatcher;
    int mN;
    int mHalfPatchSize, mIterations, mPyramids;
    bool mbHasGyroPredictInitial, mbInverse, mbConsiderIllumination, mbConsiderAffineDeformation;
    bool mbRegularizationPenalty, mbCalculateNCC;
    std::vector<cv::uchar> mvGyroPredictStatus;  // snapshot of mvStatus (reference :58)
};
