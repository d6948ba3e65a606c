// Audio replacement path (--stretch_audio): describealign.py:230-416 and :1135-1153 on the GPU.
// Kernels and their host orchestration live in dalign_stretch.hip; dalign_api.cpp wraps them in
// the C ABI (da_replace_segments / da_stretch_resident).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace da {

struct StretchState;                       // device scratch + the jump schedules of the last call
StretchState* stretch_create();
void stretch_destroy(StretchState* s);

struct StretchTimes {                      // device time per stage of the last call, ms (HIP events)
  double prepare_ms = 0;                   // int16 -> float16, loudness matching
  double resample_ms = 0, resample_points = 0, resample_bytes = 0;
  double correlate_ms = 0, correlate_windows = 0, correlate_bytes = 0;
  double viterbi_ms = 0;
  double splice_ms = 0, splice_points = 0;
  double finish_ms = 0;                    // peak normalisation + int16 interleave
};

// video / audio: device float16 planar, channel stride stretch_channel_stride(n).  video is modified in place.  Returns 0 or a
// negative da error code with the message in err.
int stretch_replace(StretchState* s, hipStream_t stream, uint16_t* d_video, int64_t n_video,
                    const uint16_t* d_audio, int64_t n_audio, int channels, const double* audio_times,
                    const double* video_times, int n_nodes, bool no_pitch_correction, StretchTimes& t,
                    std::string& err);

// Device float16 arrays are planar with this channel stride (elements), a multiple of 64 so that
// every channel starts 128-byte aligned.
int64_t stretch_channel_stride(int64_t n);

// describealign.py:156 + :1135-1148 on int16 PCM (planar or interleaved): float16 conversion and
// loudness matching in two streaming passes (moments, then convert + scale).  factors[channels]
// receives video_std / audio_std.
int stretch_prepare(StretchState* s, hipStream_t stream, const int16_t* d_pcm_video, int64_t n_video, int planar_video,
                    const int16_t* d_pcm_audio, int64_t n_audio, int planar_audio, int channels, uint16_t* d_video,
                    uint16_t* d_audio, double* factors, std::string& err);
// describealign.py:1153 and :136: peak normalise, convert to int16, interleave
int stretch_finish(StretchState* s, hipStream_t stream, uint16_t* d_video, int64_t n_video, int channels,
                   int16_t* d_out_interleaved, std::string& err);

// jump schedule (input index, signed distance) of the k-th stretched interval of the last call
const std::vector<int64_t>* stretch_schedule(const StretchState* s, int k);
int stretch_schedule_count(const StretchState* s);

}  // namespace da
