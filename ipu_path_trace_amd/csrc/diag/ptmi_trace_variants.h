// diag/ptmi_trace_variants.h -- profiling build only: the trace-kernel A/B switch (PTMI_TRACE_KERNEL, read per launch so
// that the rounds of scripts/ab_trace.py interleave in one process).  Returns false when no variant is selected: the
// caller launches the product kernel.  Part of the translation unit ptmi.hip.
#pragma once

static bool launch_trace_variant(pt_handle h, const ptd::TraceParams& P, const TraceGrid& g) {
  const char* tk = getenv("PTMI_TRACE_KERNEL");
  if (!tk) return false;
  const dim3 grid(g.blocks), block(ptd::kTraceBlock);
  hipStream_t st = h->trace_stream;
  if (!strcmp(tk, "opt0")) hipLaunchKernelGGL(ptd::trace_kernel_opt<0>, grid, block, 0, st, P);          // the round-3 kernel: neither round-4 change
  else if (!strcmp(tk, "opt1")) hipLaunchKernelGGL(ptd::trace_kernel_opt<1>, grid, block, 0, st, P);     // + reciprocal index split
  else if (!strcmp(tk, "opt2")) hipLaunchKernelGGL(ptd::trace_kernel_opt<2>, grid, block, 0, st, P);     // + camera-ray constants
  else if (!strcmp(tk, "pipe")) hipLaunchKernelGGL(ptd::trace_kernel_opt<259>, grid, block, 0, st, P);   // the next object's constants requested one object ahead
  else if (!strcmp(tk, "scenec")) hipLaunchKernelGGL(ptd::trace_kernel_opt<131>, grid, block, 0, st, P); // the object loop unrolled over the compile-time scene (pt_trace_scene_c.h)
  else if (!strcmp(tk, "fn3")) hipLaunchKernelGGL(ptd::trace_kernel_opt<67>, grid, block, 0, st, P);     // the product's structure with round 3's intersect / shading functions (pt_trace_r3fn.h)
  else if (!strcmp(tk, "rounds")) hipLaunchKernelGGL(ptd::trace_kernel_opt<35>, grid, block, 0, st, P);  // the secondary phase in material-sorted workgroup rounds (pt_trace_rounds.h)
  else if (!strcmp(tk, "cut2")) hipLaunchKernelGGL(ptd::trace_kernel_opt<7>, grid, block, 0, st, P);     // timing only: no secondary phase
  else if (!strcmp(tk, "cut1")) hipLaunchKernelGGL(ptd::trace_kernel_opt<11>, grid, block, 0, st, P);    // timing only: primary phase alone
  else if (!strcmp(tk, "count")) {   // secondary-phase occupancy counters into the stamp buffer (pt_diag_stamps)
    if (!h->d_stamps) {
      if (hipMalloc(reinterpret_cast<void**>(&h->d_stamps), 256 * 8) != hipSuccess) return false;
      (void)hipMemset(h->d_stamps, 0, 256 * 8);
    }
    ptd::TraceParams PC = P;
    PC.diag = h->d_stamps;
    hipLaunchKernelGGL(ptd::trace_kernel_opt<19>, grid, block, 0, st, PC);
  } else return false;
  return true;
}
