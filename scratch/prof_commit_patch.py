"""Development: patches the ISX_FLOOD_PROF timers to cover only the committing (or only the blocked) outer iterations."""
import sys
mode = sys.argv[1] if len(sys.argv) > 1 else "commit"
p='/root/repo/iterseg_amd/csrc/ws_flood_window.h'; s=open(p).read()
s=s.replace("    unsigned long long tk_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk_prev = __builtin_readcyclecounter();\n#define WTICK(i) do { asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); const unsigned long long t_ = __builtin_readcyclecounter(); tk_acc[i] += t_ - tk_prev; tk_prev = t_; } while (0)",
"    unsigned long long tk_acc[16] = {0}, tk_cur[16] = {0}, tk_prev = __builtin_readcyclecounter(); unsigned long long n_sel = 0; bool sel = false;\n#define WTICK(i) do { asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); const unsigned long long t_ = __builtin_readcyclecounter(); tk_cur[i] += t_ - tk_prev; tk_prev = t_; } while (0)")
old="        WTICK(8);\n"
new="        WTICK(8);\n#ifdef ISX_FLOOD_PROF\n        if (sel) { for (int i = 0; i < 16; ++i) tk_acc[i] += tk_cur[i]; ++n_sel; }\n        for (int i = 0; i < 16; ++i) tk_cur[i] = 0;\n#endif\n"
assert old in s; s=s.replace(old,new,1)
old="        for (int burst = 0; front_ready && burst"
new="#ifdef ISX_FLOOD_PROF\n        sel = %sfront_ready;\n#endif\n        for (int burst = 0; front_ready && burst" % ("" if mode == "commit" else "!")
assert old in s; s=s.replace(old,new,1)
s=s.replace("        for (int i = 0; i < 10; ++i) { atomicAdd(&stats[32 + i], tk_acc[i]);","        atomicAdd(&stats[43], n_sel);\n        for (int i = 0; i < 10; ++i) { atomicAdd(&stats[32 + i], tk_acc[i]);")
s=s.replace("        for (int i = 0; i < 10; ++i) { atomicAdd(&stats[32 + i], tk_acc[i]);","        for (int i = 10; i < 16; ++i) atomicAdd(&stats[10 + i], tk_acc[i]);\n        for (int i = 0; i < 10; ++i) { atomicAdd(&stats[32 + i], tk_acc[i]);")
s=s.replace("            WIN_FENCE();\n        for (unsigned long long todo = __ballot(has && !tocold); todo; todo &= todo - 1ull) {","            WIN_FENCE();\n        WTICK(10);\n        for (unsigned long long todo = __ballot(has && !tocold); todo; todo &= todo - 1ull) {")
s=s.replace("        if (nstage >= 64 && !serial) flush_stage(64);\n    };","        WTICK(11);\n        if (nstage >= 64 && !serial) flush_stage(64);\n        WTICK(12);\n    };")
s=s.replace("                if (cold.count > 0) { scount = cold_drain(cold, srun, hist, lane, drain_n); shead = 0; WSTAT(ST_DRAIN, 1); WCHECK(40); }\n            }\n            while (nw < wcap && scount > 0) {","                if (cold.count > 0) { scount = cold_drain(cold, srun, hist, lane, drain_n); shead = 0; WSTAT(ST_DRAIN, 1); WCHECK(40); }\n            }\n            WTICK(13);\n            while (nw < wcap && scount > 0) {")
s=s.replace("                    if (cold.count > 0) { scount = cold_drain(cold, srun, hist, lane, drain_n); shead = 0; WSTAT(ST_DRAIN, 1); WCHECK(41); }","                    WTICK(14);\n                    if (cold.count > 0) { scount = cold_drain(cold, srun, hist, lane, drain_n); shead = 0; WSTAT(ST_DRAIN, 1); WCHECK(41); }\n                    WTICK(13);")
open(p,'w').write(s)
p='/root/repo/iterseg_amd/csrc/ws.hip'; s=open(p).read()
s=s.replace("""                (double)hs[32] / hs[0], (double)hs[33] / hs[0], (double)hs[34] / hs[0], (double)hs[35] / hs[0], (double)hs[36] / hs[0],
                (double)hs[37] / hs[0], (double)hs[38] / hs[0], (double)hs[39] / hs[0], (double)hs[40] / hs[0]);""","""                (double)hs[32] / hs[43] / 64, (double)hs[33] / hs[43] / 64, (double)hs[34] / hs[43] / 64, (double)hs[35] / hs[43] / 64, (double)hs[36] / hs[43] / 64,
                (double)hs[37] / hs[43] / 64, (double)hs[38] / hs[43] / 64, (double)hs[39] / hs[43] / 64, (double)hs[40] / hs[43] / 64);
        fprintf(stderr, "[isx flood/window] EXPERIMENT: the line above is for %llu selected outer iterations only (cold tk /64: %llu %llu %llu %llu)\\n   enqueue: tier+stage %.0f hot loop %.0f flush %.0f | refill: drains %.0f window fill %.0f + %.0f\\n", hs[43], hs[44] / 64, hs[45] / 64, hs[46] / 64, hs[47] / 64, (double)hs[20] / hs[43] / 64, (double)hs[21] / hs[43] / 64, (double)hs[22] / hs[43] / 64, (double)hs[23] / hs[43] / 64, (double)hs[24] / hs[43] / 64, (double)hs[40] / hs[43] / 64);""")
open(p,'w').write(s)
