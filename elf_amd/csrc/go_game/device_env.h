// device_env.h -- C++ convenience layer of the GameContext host code over the C-ABI of
// libgoenv.so (include/goenv.h).  No HIP headers here: the host side talks to the GPU only
// through that ABI.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/goenv.h"

namespace gogame {

inline void chk(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + goenv_last_error());
}

// Device buffer owned by the host code (staging for host-resident batch tensors, index lists).
class DevBuf {
 public:
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void ensure(goenv_ctx* ctx, size_t bytes) {
    if (bytes <= cap_) return;
    release();
    chk(goenv_device_alloc(ctx, bytes, &p_), "goenv_device_alloc");
    cap_ = bytes;
    ctx_ = ctx;
  }
  // must run before the context it was allocated under is destroyed
  void release() {
    if (p_) goenv_device_free(ctx_, p_);
    p_ = nullptr;
    cap_ = 0;
  }
  void* get() const { return p_; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p_); }

 private:
  void* p_ = nullptr;
  size_t cap_ = 0;
  goenv_ctx* ctx_ = nullptr;
};

// Pinned host memory that kernels read and write IN PLACE (hipHostMalloc: one address for host and device): small index and
// result arrays of a batch travel without a copy call.  The owner orders its own reuse (a stream synchronise between a
// kernel that uses the block and the host's next write / read of it).
class PinBuf {
 public:
  PinBuf() = default;
  PinBuf(const PinBuf&) = delete;
  PinBuf& operator=(const PinBuf&) = delete;
  ~PinBuf() { release(); }
  void ensure(size_t bytes) {
    if (bytes <= cap_) return;
    release();
    const size_t cap = bytes < 4096 ? 4096 : bytes + bytes / 2;
    chk(goenv_host_alloc(cap, &p_), "goenv_host_alloc");
    cap_ = cap;
  }
  void release() {
    if (p_) goenv_host_free(p_);
    p_ = nullptr;
    cap_ = 0;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p_); }

 private:
  void* p_ = nullptr;
  size_t cap_ = 0;
};

// The boards of one GameContext on one GPU.
class DeviceEnv {
 public:
  // `scratch` = 1 adds one board behind the context's own (index n()) for count_legal_prefix
  DeviceEnv(int device, int n_boards, uint64_t seed, int64_t id_base, float komi, int max_ply, int scratch = 0)
      : n_(n_boards), scratch_(scratch ? n_boards : -1) {
    chk(goenv_create(device, n_boards + (scratch ? 1 : 0), seed, id_base, komi, max_ply, &ctx_), "goenv_create");
  }
  DeviceEnv(const DeviceEnv&) = delete;
  DeviceEnv& operator=(const DeviceEnv&) = delete;
  ~DeviceEnv() {
    one_i32_.release(); one_i32b_.release(); one_i64_.release(); moves_.release();
    if (ctx_) goenv_destroy(ctx_);
  }

  goenv_ctx* ctx() const { return ctx_; }
  int n() const { return n_; }
  void set_stream(void* s) { stream_ = s; }
  void* stream() const { return stream_; }
  void sync() { chk(goenv_sync(ctx_, stream_), "goenv_sync"); }
  void h2d(void* dst, const void* src, size_t bytes) { chk(goenv_copy(ctx_, dst, src, bytes, stream_), "goenv_copy"); }
  void d2h(void* dst, const void* src, size_t bytes) { chk(goenv_copy(ctx_, dst, src, bytes, stream_), "goenv_copy"); }

  // --- synchronous single-board operations (the offline loader's rare paths) ---------------

  // GoState::Reset + ApplyHandicap (go/go_state.cc:85-91)
  void reset_one(int board, int handicap) {
    one_i32_.ensure(ctx_, 4); one_i32b_.ensure(ctx_, 4);
    const int32_t b = board, h = handicap;
    h2d(one_i32_.get(), &b, 4);
    h2d(one_i32b_.get(), &h, 4);
    chk(goenv_reset(ctx_, one_i32_.as<int32_t>(), 1, one_i32b_.as<int32_t>(), stream_), "goenv_reset");
  }

  // GoState::forward for one board; returns legality
  bool forward_one(int board, int action) {
    one_i64_.ensure(ctx_, 8); one_i32_.ensure(ctx_, 4);
    const int64_t a = action;
    h2d(one_i64_.get(), &a, 8);
    chk(goenv_ply_sel(ctx_, nullptr, board, 1, 0u, one_i64_.as<int64_t>(), nullptr, nullptr, nullptr, nullptr,
                      nullptr, one_i32_.as<int32_t>(), stream_), "goenv_ply_sel");
    int32_t st = 0;
    d2h(&st, one_i32_.get(), 4);
    sync();
    return st == 1;
  }

  // Replays `n` actions on one board in one launch; returns how many were applied before the
  // first illegal one.
  int replay_one(int board, const int16_t* moves, int n) {
    if (n <= 0) return 0;
    moves_.ensure(ctx_, (size_t)n * 2); one_i32_.ensure(ctx_, 4); one_i32b_.ensure(ctx_, 4);
    const int32_t cnt = n;
    h2d(moves_.get(), moves, (size_t)n * 2);
    h2d(one_i32_.get(), &cnt, 4);
    chk(goenv_replay(ctx_, nullptr, board, 1, moves_.as<int16_t>(), n, one_i32_.as<int32_t>(),
                     one_i32b_.as<int32_t>(), stream_), "goenv_replay");
    int32_t applied = 0;
    d2h(&applied, one_i32b_.get(), 4);
    sync();
    return applied;
  }

  // How many of `moves` GoState::forward accepts when they are replayed from an empty board with
  // `handicap` stones: one reset + one replay launch on the scratch board, synchronous.  Run once per
  // SGF record, when a loader first uses it.
  int count_legal_prefix(int handicap, const std::vector<int16_t>& moves) {
    if (scratch_ < 0) throw std::runtime_error("count_legal_prefix: the context has no scratch board");
    reset_one(scratch_, handicap);
    return replay_one(scratch_, moves.data(), (int)moves.size());
  }

 private:
  goenv_ctx* ctx_ = nullptr;
  int n_ = 0;
  int scratch_ = -1;
  void* stream_ = nullptr;
  DevBuf one_i32_, one_i32b_, one_i64_, moves_;
};

}  // namespace gogame
