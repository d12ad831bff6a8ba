// gm_pipe.cpp -- gpumap_pipe: the chunk loop of mapper() (src/mapper.c:23-40) as a pipeline over batch objects.
// Every slot owns a batch (device buffers, scratch, page-locked result arena) and a worker thread that takes a
// chunk through stage -> run (encode + search + locate + order) -> fetch.  The slots of one context share three
// streams: uploads, kernels, result copies.  The kernels of consecutive chunks so run back to back on one
// stream, never idle as long as two chunks are in their run (a worker only waits for events of its own
// chunk), while the results of the chunk before and the letters of the chunk after cross the link on the
// other two streams.  Chunks go round the slots, slots go round the contexts (GPUs), results come back in
// submit order.
// Host code only: the device work is behind gm_batch_fill / gm_batch_run / gm_batch_fetch (gm_device.hip).
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include "gm_host.hpp"

namespace {
enum SlotState { SLOT_FREE, SLOT_QUEUED, SLOT_DONE, SLOT_RETURNED };
struct Slot {
   gpumap_ctx *ctx = nullptr;
   gpumap_batch *batch = nullptr;
   std::thread worker;
   SlotState state = SLOT_FREE;
   const void *src = nullptr;
   const int64_t *offsets = nullptr;
   int64_t n_reads = 0;
   void *user = nullptr;
   gpumap_hits hits;
   int rc = 0;
   int ctx_index = 0;            // which of the pipe's contexts the slot works on
   uint64_t run_seq = 0;         // its chunk's place among the chunks of that context
};
}  // namespace

struct gpumap_pipe {
   std::vector<Slot> slots;
   int tau = 0, src_flags = 0, result_flags = 0;
   uint64_t submitted = 0, returned = 0;
   bool closed = false, stop = false;
   // The runs of the chunks of one context take turns, in submit order: the kernels of a chunk are enqueued stage by
   // stage (the host reads counts in between), and with several chunks in their run at once the stages interleave on the
   // context's kernel stream: every chunk then takes as long as all of them together, in lock-step (results came back in
   // bursts of `depth`, the first after 58 ms instead of 18).  Uploads and result copies still overlap the runs.
   std::vector<uint64_t> ctx_submitted, ctx_turn;
   std::vector<uint64_t> ctx_ordering;      // runs of the context whose sort has been enqueued (or that ended without one)
   bool ordered = true, overlap = true, gate = true;
   // the last thread that took or released a result (gpumap_pipe_submit: can anyone but the submitting thread free a slot?)
   bool consumer_seen = false;
   std::thread::id consumer_tid;
   int grace_ms = 2000;
   std::mutex m;
   std::condition_variable cv;
};

static void pipe_worker(gpumap_pipe *p, Slot *s)
{
   for (;;) {
      {
         std::unique_lock<std::mutex> l(p->m);
         p->cv.wait(l, [&] { return s->state == SLOT_QUEUED || p->stop; });
         if (p->stop) return;
      }
      memset(&s->hits, 0, sizeof s->hits);
      static const bool dbg = getenv("GPUMAP_PIPE_DEBUG") != nullptr;
      const int id = (int)(s - p->slots.data());
      auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
      if (dbg) fprintf(stderr, "[pipe %.2f] slot %d: fill %ld reads\n", now_ms(), id, (long)s->n_reads);
      int rc = gm_batch_fill(s->ctx, s->batch, s->src, s->offsets, s->n_reads, p->src_flags, false);
      if (dbg) fprintf(stderr, "[pipe %.2f] slot %d: run (rc %d)\n", now_ms(), id, rc);
      if (p->ordered) {
         std::unique_lock<std::mutex> l(p->m);
         p->cv.wait(l, [&] { return p->ctx_turn[(size_t)s->ctx_index] == s->run_seq; });
      }
      // The turn passes on when the search of this run is over: what remains of it (locate, order, hit lines) runs on
      // the post stream beside the next chunk's encoding and search (GPUMAP_PIPE_OVERLAP=0: when the whole run is over).
      bool passed = false;
      auto pass_turn = [&] {
         if (!p->ordered || passed) return;
         passed = true;
         { std::lock_guard<std::mutex> l(p->m); p->ctx_turn[(size_t)s->ctx_index]++; }
         p->cv.notify_all();
      };
      // The k_search of this run is not launched before the sorts of the runs before it have been enqueued (see gm_batch_set_gate).
      auto ordering_enqueued = [&] {
         { std::lock_guard<std::mutex> l(p->m); uint64_t &k = p->ctx_ordering[(size_t)s->ctx_index]; k = std::max(k, s->run_seq + 1); }
         p->cv.notify_all();
      };
      auto wait_for_ordering = [&] {
         std::unique_lock<std::mutex> l(p->m);
         p->cv.wait(l, [&] { return p->ctx_ordering[(size_t)s->ctx_index] >= s->run_seq || p->stop; });
      };
      if (p->overlap) gm_batch_set_hook(s->batch, pass_turn);
      if (p->overlap && p->ordered && p->gate) gm_batch_set_gate(s->batch, wait_for_ordering, ordering_enqueued);
      if (rc == 0) rc = gm_batch_run(s->ctx, s->batch, p->tau, 0, p->result_flags);
      gm_batch_set_hook(s->batch, nullptr);
      gm_batch_set_gate(s->batch, nullptr, nullptr);
      ordering_enqueued();
      pass_turn();
      if (dbg) fprintf(stderr, "[pipe %.2f] slot %d: fetch (rc %d)\n", now_ms(), id, rc);
      if (rc == 0) rc = gm_batch_fetch(s->ctx, s->batch, &s->hits, p->result_flags);
      if (dbg) fprintf(stderr, "[pipe %.2f] slot %d: done (rc %d)\n", now_ms(), id, rc);
      {
         std::lock_guard<std::mutex> l(p->m);
         s->rc = rc;
         s->state = SLOT_DONE;
      }
      p->cv.notify_all();
   }
}

extern "C" void gpumap_pipe_destroy(gpumap_pipe *p)
{
   if (!p) return;
   {
      std::unique_lock<std::mutex> l(p->m);
      // chunks in flight are finished first: their source buffers and the batch objects are in use
      p->cv.wait(l, [&] { for (auto &s : p->slots) if (s.state == SLOT_QUEUED) return false; return true; });
      p->stop = true;
   }
   p->cv.notify_all();
   for (auto &s : p->slots) {
      if (s.worker.joinable()) s.worker.join();
      if (s.batch) gm_ctx_give_batch(s.ctx, s.batch);
   }
   delete p;
}

extern "C" gpumap_pipe *gpumap_pipe_create(gpumap_ctx *const *ctxs, int n_ctx, int depth, int tau, int flags)
{
   gpumap_pipe *p = nullptr;
   GM_FAIL_IF(ctxs == nullptr || n_ctx < 1, "no context.");
   for (int i = 0; i < n_ctx; i++) GM_FAIL_IF(ctxs[i] == nullptr, "a context is NULL.");
   GM_FAIL_IF(depth < 1 || depth > 16, "depth out of range (1..16).");
   GM_FAIL_IF(tau < 0 || tau > GPUMAP_MAX_TAU, "argument 'tau' out of range.");
   GM_FAIL_IF(flags & ~(GPUMAP_SRC_TEXT | GPUMAP_SRC_DEVICE | GPUMAP_SRC_FILE | GPUMAP_PIPE_COMPACT | GPUMAP_PIPE_LINES | GPUMAP_PIPE_PACKED), "unknown flag.");
   GM_FAIL_IF((flags & GPUMAP_PIPE_PACKED) && (flags & GPUMAP_PIPE_LINES), "packed results or hit lines, not both.");
   GM_FAIL_IF((flags & GPUMAP_SRC_DEVICE) && n_ctx > 1, "device-resident chunks belong to one GPU: one context.");
   GM_FAIL_IF((flags & GPUMAP_SRC_FILE) && (flags & GPUMAP_SRC_DEVICE), "blocks of a read file come from host memory.");
   GM_FAIL_IF((flags & GPUMAP_PIPE_LINES) && !(flags & GPUMAP_SRC_FILE), "hit lines need the blocks of the read file (GPUMAP_SRC_FILE: the tags).");
   p = new gpumap_pipe();
   p->tau = tau;
   p->src_flags = flags & (GPUMAP_SRC_TEXT | GPUMAP_SRC_DEVICE | GPUMAP_SRC_FILE);
   p->result_flags = GM_RESULT_PINNED | GM_RESULT_ONE_ARENA |
                     ((flags & GPUMAP_PIPE_LINES) ? GM_RESULT_LINES : (flags & GPUMAP_PIPE_PACKED) ? (GM_RESULT_COMPACT | GM_RESULT_PACKED)
                                                                     : (flags & GPUMAP_PIPE_COMPACT) ? GM_RESULT_COMPACT : 0);
   p->slots = std::vector<Slot>((size_t)depth * (size_t)n_ctx);
   p->ctx_submitted.assign((size_t)n_ctx, 0);
   p->ctx_turn.assign((size_t)n_ctx, 0);
   p->ctx_ordering.assign((size_t)n_ctx, 0);
   p->gate = !getenv("GPUMAP_PIPE_GATE") || atoi(getenv("GPUMAP_PIPE_GATE")) != 0;      // (0: tuning aid)
   if (getenv("GPUMAP_PIPE_GRACE_MS")) p->grace_ms = std::max(0, atoi(getenv("GPUMAP_PIPE_GRACE_MS")));
   p->overlap = !getenv("GPUMAP_PIPE_OVERLAP") || atoi(getenv("GPUMAP_PIPE_OVERLAP")) != 0;
   p->ordered = !getenv("GPUMAP_PIPE_ORDERED") || atoi(getenv("GPUMAP_PIPE_ORDERED")) != 0;      // (0: tuning aid, the runs interleave)
   for (size_t i = 0; i < p->slots.size(); i++) {
      Slot &s = p->slots[i];
      s.ctx = ctxs[i % (size_t)n_ctx];                   // consecutive chunks go to different GPUs
      s.ctx_index = (int)(i % (size_t)n_ctx);
      s.batch = gm_ctx_take_batch(s.ctx);
      GM_FAIL_IF(s.batch == nullptr, "cannot create a batch.");
      void *run = nullptr, *copy = nullptr, *up = nullptr, *post = nullptr;
      GM_FAIL_IF(gm_ctx_pipe_streams(s.ctx, &run, &copy, &up, &post) != 0, "cannot create the streams of the pipeline.");
      gm_batch_set_streams(s.batch, run, copy, up, p->overlap ? post : run);
   }
   for (auto &s : p->slots) s.worker = std::thread(pipe_worker, p, &s);
   return p;
failure_return:
   gpumap_pipe_destroy(p);
   return nullptr;
}

static int pipe_submit(gpumap_pipe *p, const void *src, const int64_t *offsets, int64_t n_reads, void *user, bool may_block)
{
   bool full = false;
   GM_FAIL_IF(p == nullptr, "argument 'pipe' is NULL.");
   GM_FAIL_IF(offsets == nullptr || n_reads < 0, "invalid arguments.");
   {
      std::unique_lock<std::mutex> l(p->m);
      GM_FAIL_IF(p->closed, "the pipe is closed.");
      // The slot of this chunk is the one the counter names once the wait is over (several threads may submit).
      auto slot_free = [&] { return p->slots[p->submitted % p->slots.size()].state == SLOT_FREE; };
      if (!slot_free()) {
         if (!may_block) { full = true; goto failure_return; }      // gpumap_pipe_try_submit: not an error
         // A slot is freed by gpumap_pipe_next (a failed chunk) or gpumap_pipe_release, which only the caller's threads
         // call.  With every slot taken, a caller whose results are taken by this same thread (mapper() is single-threaded,
         // src/mapper.c:23-40) would wait here for ever: unless another thread has been seen taking results (one that is
         // just starting gets a moment), the call fails instead.
         const std::thread::id me = std::this_thread::get_id();
         auto other = [&] { return p->consumer_seen && p->consumer_tid != me; };
         if (!other()) {
            p->cv.wait_for(l, std::chrono::milliseconds(p->grace_ms), [&] { return slot_free() || other() || p->closed; });
            GM_FAIL_IF(!slot_free() && !other(),
                       "every slot of the pipe holds a chunk and no other thread takes results: call gpumap_pipe_next / gpumap_pipe_release before submitting more (or use gpumap_pipe_try_submit).");
         }
         p->cv.wait(l, [&] { return slot_free() || p->closed; });
         GM_FAIL_IF(!slot_free(), "the pipe is closed.");
      }
      Slot &s = p->slots[p->submitted % p->slots.size()];
      s.src = src; s.offsets = offsets; s.n_reads = n_reads; s.user = user;
      s.run_seq = p->ctx_submitted[(size_t)s.ctx_index]++;
      s.state = SLOT_QUEUED;
      p->submitted++;
   }
   p->cv.notify_all();
   return 1;
failure_return:
   return full ? 0 : -1;
}

extern "C" int gpumap_pipe_submit(gpumap_pipe *p, const void *src, const int64_t *offsets, int64_t n_reads, void *user)
{
   return pipe_submit(p, src, offsets, n_reads, user, true) == 1 ? 0 : -1;
}

// 1: submitted; 0: every slot holds a chunk (take a result first); -1: error.  Never waits.
extern "C" int gpumap_pipe_try_submit(gpumap_pipe *p, const void *src, const int64_t *offsets, int64_t n_reads, void *user)
{
   return pipe_submit(p, src, offsets, n_reads, user, false);
}

extern "C" int64_t gpumap_pipe_in_flight(gpumap_pipe *p)
{
   if (!p) return 0;
   std::lock_guard<std::mutex> l(p->m);
   return (int64_t)(p->submitted - p->returned);
}

extern "C" void gpumap_pipe_close(gpumap_pipe *p)
{
   if (!p) return;
   { std::lock_guard<std::mutex> l(p->m); p->closed = true; }
   p->cv.notify_all();
}

extern "C" int gpumap_pipe_next(gpumap_pipe *p, gpumap_hits *out, void **user)
{
   GM_FAIL_IF(p == nullptr || out == nullptr, "NULL argument.");
   {
      std::unique_lock<std::mutex> l(p->m);
      p->consumer_seen = true; p->consumer_tid = std::this_thread::get_id();
      p->cv.notify_all();
      // a caller with a thread of its own for the results waits here for the submitting thread; a single-threaded
      // caller only asks when it has something in flight
      p->cv.wait(l, [&] { return p->returned < p->submitted || p->closed; });
      if (p->returned == p->submitted) return 0;
      Slot &s = p->slots[p->returned % p->slots.size()];
      p->cv.wait(l, [&] { return s.state == SLOT_DONE; });
      p->returned++;
      if (user) *user = s.user;
      if (s.rc != 0) {
         memset(out, 0, sizeof *out);
         s.state = SLOT_FREE;
         l.unlock();
         p->cv.notify_all();
         GM_FAIL("a chunk failed in the pipeline.");
      }
      *out = s.hits;
      s.state = SLOT_RETURNED;
   }
   return 1;
failure_return:
   return -1;
}

extern "C" void gpumap_pipe_release(gpumap_pipe *p, gpumap_hits *hits)
{
   if (!p || !hits) return;
   {
      std::lock_guard<std::mutex> l(p->m);
      p->consumer_seen = true; p->consumer_tid = std::this_thread::get_id();
      for (auto &s : p->slots)
         if (s.state == SLOT_RETURNED && s.hits.arena == hits->arena) { s.state = SLOT_FREE; break; }      // (a slot's arena is its own)
   }
   const bool expanded = hits->expanded != 0;
   if (expanded) gpumap_hits_free(hits);            // (frees what gpumap_hits_expand added; the arena is the pipe's)
   else memset(hits, 0, sizeof *hits);
   p->cv.notify_all();
}
