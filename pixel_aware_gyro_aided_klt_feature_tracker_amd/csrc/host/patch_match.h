// patch_match.h -- the reference's PatchMatch class (include/patch_match.h:41-103): same constructor, same public
// methods, bodies replaced by the MI355X path behind the C ABI (pagk.h).
#pragma once
#include <cmath>
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

    // reference :51, src/patch_match.cpp:61-76.  Level 0 = the tracker's two images (shared headers), level i = the
    // exact-2x decimation built ON THE DEVICE (pagk_frame_upload: both frames go to this thread's context, slots 0 / 1)
    // and read back, so that mvImgPyr1 / mvImgPyr2 hold what the kernels sample; mvScales as :66,:73.
    void CreatePyramids();

    // Multi level optical flow tracking (reference :54, src/patch_match.cpp:79-142): gathers the tracker's inputs, runs
    // pagk_track (CreatePyramids + every level + DistortPoints + SetMatcher in the library), scatters into the six result
    // vectors (:370-388) and leaves the members below as the reference's would be.
    // Throws std::runtime_error when the HIP path cannot run: there is no CPU fallback.
    void OpticalFlowMultiLevel();

    // reference :57-60, src/patch_match.cpp:167-367: ONE feature at the current level (mLevel) -- reads mvPtPyr1Un[i] /
    // mvPtPyr2Un[i], writes mvPtPyr2Un[i] and, at level 0, mvSuccess[i] / mvPixelErrorsOfPatchMatched[i]; mvNcc[i] at every
    // level (:356-366).  One single-level launch of the library on the level's images (pagk_track_pyr).  Needs
    // CreatePyramids() first; the point vectors are initialised on first use as :83-90 does.
    void OpticalFlowConsideringIlluminationChange_onePixel(const int i, const bool bConsiderIllumination,
                                                           const bool bConsiderAffineDeformation,
                                                           const bool bRegularizationPenalty);
    // reference :61, src/patch_match.cpp:370-388: the six result vectors of the tracker from the members below
    void SetMatcher();

    // reference :64, src/patch_match.cpp:391-406 (the MEMBER sampler: `>=` clamps, factored formula); bytes past the image
    // buffer and row padding read as 0, as in the library
    inline float GetPixelValue(const cv::Mat &img, float x, float y) const;

    // reference :66, src/patch_match.cpp:409-416 -> DistortVecPoints src/utils.cpp:49-76: mvPtPyr2 from mvPtPyr2Un
    void DistortPoints();

    // reference :69, src/patch_match.cpp:433-469: zero-normalised cross correlation, x outer / y inner, float sums
    float NCC(int halfPathSize, const cv::Mat &ref, const cv::Mat &cur, const cv::Point2f &pt_ref, const cv::Point2f &pt_cur,
              const cv::Mat &warp_mat);

    // ---- not in the reference ----
    // mLevel is private there and set by OpticalFlowMultiLevel's loop (:99); a caller that drives
    // OpticalFlowConsideringIlluminationChange_onePixel itself selects the level here.
    void SetLevel(int level) { mLevel = level; }
    // The context (device buffers, stream) this thread's PatchMatch instances share.
    static pagk_ctx *Context(int device = 0);
    static void ReleaseContext();

private:
    void InitPoints();              // :83-95
    pagk_params MakeParams() const; // the constructor's constants (:48-57) + the tracker's camera model

    GyroAidedTracker *mpMatcher;
    int mN;
    int mHalfPatchSize, mIterations, mPyramids;
    bool mbHasGyroPredictInitial, mbInverse, mbConsiderIllumination, mbConsiderAffineDeformation;
    bool mbRegularizationPenalty, mbCalculateNCC;
    double mPyramidScale;  // :54
    int mLevel;
    std::vector<float> mvScales;
    std::vector<bool> mvSuccess;
    std::vector<double> mvPixelErrorsOfPatchMatched;
    std::vector<float> mvNcc;
    std::vector<cv::uchar> mvGyroPredictStatus;  // snapshot of mvStatus (reference :58)
    std::vector<cv::Mat> mvImgPyr1, mvImgPyr2;
    std::vector<cv::Point2f> mvPtPyr1Un, mvPtPyr2, mvPtPyr2Un;
};

inline float PatchMatch::GetPixelValue(const cv::Mat &img, float x, float y) const
{
    if (x < 0) x = 0;  // :394-397
    if (y < 0) y = 0;
    if (x >= img.cols) x = img.cols - 1;
    if (y >= img.rows) y = img.rows - 1;
    const size_t off = (size_t)int(y) * img.step + (size_t)int(x);  // :399, linear addressing
    auto tap = [&](size_t o) -> float {
        const size_t r = o / img.step, c = o % img.step;
        return (r < (size_t)img.rows && c < (size_t)img.cols) ? (float)img.data[o] : 0.0f;
    };
    const float xx = x - std::floor(x), yy = y - std::floor(y);  // :400
    const float a = 1.0f - xx, b = 1.0f - yy;
    return b * (a * tap(off) + xx * tap(off + 1)) + yy * (a * tap(off + img.step) + xx * tap(off + img.step + 1));  // :402-403
}
