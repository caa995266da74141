/* libfulgor_gpu.so — C ABI of the MI355X pseudoalignment engine.
 *
 * The reference (jermp/fulgor v4.2.0) has no FFI layer: the hot path sits behind three const member
 * functions of `template <typename ColorSets> struct index` (include/index.hpp:39-46)
 *
 *   void fetch_color_set_ids(std::string const& sequence, std::vector<uint32_t>& color_set_ids) const;
 *   void pseudoalign_full_intersection(std::vector<uint32_t>& color_set_ids,
 *                                      std::vector<uint32_t>& results, std::vector<uint32_t>& tmp) const;
 *   void pseudoalign_threshold_union(std::string const& sequence, std::vector<uint32_t>& results,
 *                                    const double threshold) const;
 *
 * called once per read from pseudoalign_worker (tools/pseudoalign.cpp:22-51). A GPU wants batches, so
 * every entry point below is the batch-oriented restatement of one of those members: reads are passed
 * as concatenated ASCII bases + (n+1) offsets, results come back as CSR (offsets[n+1] + values) with
 * every per-read list sorted ascending, exactly the vectors the members would have produced, in read
 * order. Plain pointers and sizes only; no C++ or torch types.
 *
 * Errors: every int-returning function returns 0 on success and a negative errno-style code on
 * failure; fgpu_last_error() then holds a message (thread local). The reference throws
 * std::runtime_error on load failures (include/util.hpp:91-95, src/index.cpp:65,138) and has no error
 * path on queries; a missing GPU / failed HIP call is reported here as -EIO and never falls back to
 * the CPU.
 */
#ifndef FULGOR_GPU_H
#define FULGOR_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fgpu_index fgpu_index;   /* index resident in HBM (replaces index<ColorSets>)          */
typedef struct fgpu_reads fgpu_reads;   /* a batch of reads resident in HBM                            */
typedef struct fgpu_result fgpu_result; /* CSR results of one batch, resident in HBM, reusable         */

enum { FGPU_FULL_INTERSECTION = 0, FGPU_THRESHOLD_UNION = 1 }; /* pseudoalignment_algorithm, src/ps_utils.cpp:12 */
enum { FGPU_HYBRID = 0, FGPU_DIFF = 1, FGPU_META = 2, FGPU_META_DIFF = 3 }; /* index_t, include/util.hpp:18 */

const char* fgpu_last_error(void);

/* essentials::load(index, filename) + upload (tools/pseudoalign.cpp:340). `path` is a dump basename as
 * written by `fulgor dump` (src/index.cpp:59-120) or an .fgidx container written by fgpu_save.
 * device >= 0: HIP device ordinal. device == FGPU_HOST_ONLY: ingest on the host only (for fgpu_save /
 * fgpu_info / fgpu_export / fgpu_selfcheck); every query entry point then fails with -ENODEV. */
#define FGPU_HOST_ONLY (-1)
int fgpu_open(const char* path, int device, fgpu_index** out);
/* build-time self check (the reference's `--check`, include/builders/builder.hpp:221-277): every k-mer
 * of every `unitig_stride`-th unitig must resolve to its unitig's colour-set id through the dictionary */
int fgpu_selfcheck(const fgpu_index* idx, uint64_t unitig_stride);
/* Re-encode the colour sets with another codec of the reference (index_types.hpp): FGPU_DIFF
 * (differential.hpp), FGPU_META (meta.hpp), FGPU_META_DIFF (meta_differential.hpp), FGPU_HYBRID to go back.
 * Stands in for `fulgor build --meta / --diff` + `fulgor permute` whose clustering heuristics are out of
 * scope: partitions are colour ranges of `partition_size`, clusters are runs of `cluster_size` consecutive
 * sets; colour numbering and colour-set ids are unchanged, so query results are identical across codecs.
 * Subsequent queries (and fgpu_save) use the new codec. */
int fgpu_convert(fgpu_index* idx, int index_type, uint32_t partition_size, uint32_t cluster_size);
void fgpu_close(fgpu_index* idx);
int fgpu_save(const fgpu_index* idx, const char* path);
/* index::k / num_colors / num_color_sets / num_unitigs (include/index.hpp:64-68), ColorSets::type */
int fgpu_info(const fgpu_index* idx, uint64_t* k, uint64_t* num_colors, uint64_t* num_color_sets,
              uint64_t* num_unitigs, uint64_t* num_kmers, int* index_type);

/* ---- host-buffer calls: one per reference member; outputs belong to the library (malloc'd, or slabs of its pinned pool): release them with fgpu_free ONLY ---- */
/* index::fetch_color_set_ids (src/ps_full_intersection.cpp:334-374) */
int fgpu_fetch_color_set_ids(fgpu_index* idx, const char* bases, const uint64_t* offs, uint64_t n,
                             uint64_t** out_offsets, uint32_t** out_ids);
/* fetch_color_set_ids + index::pseudoalign_full_intersection (src/ps_full_intersection.cpp:376-400),
 * i.e. the FULL_INTERSECTION arm of pseudoalign_worker (tools/pseudoalign.cpp:27-30) */
int fgpu_full_intersection(fgpu_index* idx, const char* bases, const uint64_t* offs, uint64_t n,
                           uint64_t** out_offsets, uint32_t** out_colors);
/* index::pseudoalign_threshold_union (src/ps_threshold_union.cpp:320-402) */
int fgpu_threshold_union(fgpu_index* idx, const char* bases, const uint64_t* offs, uint64_t n, double tau,
                         uint64_t** out_offsets, uint32_t** out_colors);
/* index::pseudoalign_full_intersection given the colour-set ids (the --deduplicate path feeds it this
 * way, src/ps_utils.cpp:307-415) */
int fgpu_intersect_ids(fgpu_index* idx, const uint32_t* ids, const uint64_t* id_offs, uint64_t n,
                       uint64_t** out_offsets, uint32_t** out_colors);
/* ---- k-mer level queries on the same lookup kernel (the reference's other two query tools) ----------
 * colour-set id of EVERY k-mer of every read (0xFFFFFFFF = k-mer absent or containing a non-ACGT base):
 * out_offsets[r+1]-out_offsets[r] = max(0, len_r - k + 1). Run-length encoding these ids gives
 * index::kmer_conservation's (start, num_kmers, color_set_id) triples (src/kmer_conservation.cpp:7-54); the
 * id != 0xFFFFFFFF flags are kmer_matches' positive-k-mer bit vector. */
int fgpu_kmer_color_set_ids(fgpu_index* idx, const char* bases, const uint64_t* offs, uint64_t n,
                            uint64_t** out_offsets, uint32_t** out_ids);
/* index::kmer_matches counts (src/kmer_matches.cpp:7-30): out_counts[r * num_colors + c] = number of positive
 * k-mers of read r whose colour set contains c (all zero for reads shorter than k). Dense: size n by num_colors. */
int fgpu_kmer_matches(fgpu_index* idx, const char* bases, const uint64_t* offs, uint64_t n, uint32_t** out_counts);
/* The two tools as line emitters (tools/kmer_conservation.cpp:10-56: `name <tab> #triples [<tab>(start num_kmers color_set_id)]...`;
 * tools/kmer_matches.cpp:10-57: `name <tab> #k-mers [<tab>0|1 per k-mer] [<tab>count per colour]`): a batch of records in — bases +
 * (n + 1) offsets, names + (n + 1) offsets as fgpu_fastx_names gives them —, the tool's output lines for them out, in order (malloc'd:
 * fgpu_free). Lookup and counts on the device, the text on the host's threads. An emitter is one worker of the reference reading the
 * file in order: a record shorter than k repeats what the previous record left in the worker's buffers (src/kmer_matches.cpp:11). */
typedef struct fgpu_kmer_emitter fgpu_kmer_emitter;
enum { FGPU_TOOL_KMER_CONSERVATION = 0, FGPU_TOOL_KMER_MATCHES = 1 };
int fgpu_kmer_emitter_create(fgpu_index* idx, int tool, fgpu_kmer_emitter** out);
int fgpu_kmer_emitter_add(fgpu_kmer_emitter* e, const char* bases, const uint64_t* offs, uint64_t n, const char* names, const uint64_t* name_offs,
                          char** out, uint64_t* out_len);
/* the same, the lines written to the file descriptor out_fd (no copy of the text: kmer-matches writes one count per colour and record) */
int fgpu_kmer_emitter_write(fgpu_kmer_emitter* e, const char* bases, const uint64_t* offs, uint64_t n, const char* names, const uint64_t* name_offs,
                            int out_fd, uint64_t* out_len);
void fgpu_kmer_emitter_free(fgpu_kmer_emitter* e);
void fgpu_free(void* p);

/* ---- device-resident calls (what the driver loop and bench.py use) -------------------------------- */
int fgpu_reads_upload(fgpu_index* idx, const char* bases, const uint64_t* offs, uint64_t n, fgpu_reads** out);
void fgpu_reads_free(fgpu_reads* reads);
int fgpu_result_create(fgpu_index* idx, fgpu_result** out);
void fgpu_result_free(fgpu_result* res);
/* one pass of the hot path over reads [first, first+count) of an uploaded batch; results replace the
 * previous contents of `res`. Returns after the kernels have completed. A pass leaves every result as a row of colour bits (or, up
 * to 16 colours, as the colours) plus its size: what the counters and the compressed formatter need (src/ps_utils.cpp:168-237).
 * The u32 colour lists — the `colors` vector of tools/pseudoalign.cpp:27-36 — are materialised when a consumer asks for them:
 * fgpu_result_expand, fgpu_result_download, ascii / binary formatting and the host-buffer calls above. */
int fgpu_run(fgpu_index* idx, const fgpu_reads* reads, uint64_t first, uint64_t count, int algo, double tau,
             fgpu_result* res);
/* The same pass in its two halves, for a worker loop that keeps two results in flight (pseudoalign_worker's loop over batches,
 * tools/pseudoalign.cpp:22-51, one batch ahead): fgpu_run_lookup queues fetch_color_set_ids for the reads (k-mers -> colour-set ids)
 * and returns at once; fgpu_run_colours runs the colour stage on the ids the result holds and returns after its kernels have
 * completed. fgpu_run_lookup(batch i + 1, result B) followed by fgpu_run_colours(result A of batch i) lets the lookup of one
 * batch run beside the colour stage of the batch before it; with FULGOR_CU_SPLIT=<n> in the environment when the results are
 * created the two run on disjoint parts of the device (the lookup kernels on CUs [0, n), the colour kernels on the others; a measurement
 * knob: every kernel of the pass scales with the CUs it gets). The reads and the result must stay alive until
 * fgpu_run_colours (or any other call that waits for the result's stream) has returned (fgpu_reads_free itself waits for the
 * lookups queued on the reads). */
int fgpu_run_lookup(fgpu_index* idx, const fgpu_reads* reads, uint64_t first, uint64_t count, fgpu_result* res);
int fgpu_run_colours(fgpu_index* idx, int algo, double tau, fgpu_result* res);
/* materialises the CSR colour lists of the last pass on the device now (idempotent; the per-colour hit histogram of the pass is
 * taken along). bench.py calls it inside the timed region: the metric is quoted on passes that end in u32 colour lists. */
int fgpu_result_expand(fgpu_result* res);
/* ps_options counters (src/ps_utils.cpp:417-448): reads processed / reads with a non-empty result */
int fgpu_result_sizes(const fgpu_result* res, uint64_t* num_reads, uint64_t* total_colors, uint64_t* num_mapped);
int fgpu_result_download(const fgpu_result* res, uint64_t* offsets /* n+1 */, uint32_t* colors /* total */);
/* adds this result's per-colour hit counts (#reads whose result contains colour c) followed by
 * {num_reads, num_mapped} into a DEVICE array of num_colors+2 uint64 (the vector RCCL all-reduces) */
int fgpu_result_accumulate_hits(fgpu_index* idx, const fgpu_result* res, void* device_u64_hits);
/* Verification aid in the spirit of util::check_intersection / check_union (include/util.hpp:106-208): a checksum of the u32 colour
 * lists of the last pass (materialised first if they were not), taken twice on the device by kernels that share nothing —
 * from_lists from the CSR itself (entry p holds colour c: v = (c + 1)(p + 1); {#entries, sum of v mod 2^64, xor of v * odd constant}),
 * from_rows from what the colour stage left (result rows / small-result slots, sizes, CSR offsets). Equal triples: the expansion
 * kernel wrote every colour of every read at its place. Three uint64 each. */
int fgpu_result_checksum(fgpu_result* res, uint64_t* from_lists, uint64_t* from_rows);
/* algorithmic bytes of the last run. Colour-intersection stage, per read:
 *   list side   = sum over its colour-set ids of ceil(list bits / 8) + 16 (two offsets) + 4 (the id)
 *   output side = 4 * |result| + 8 (CSR offset)
 * lookup stage = ceil(bases / 4) + 8 per k-mer
 * (meta / differential codecs on their own kernels: the list side counts every partial list the set's ops touch; on the dense rows
 * every codec is charged the hybrid lists, which is what the rows were built from) */
int fgpu_result_algorithmic_bytes(const fgpu_result* res, uint64_t* list_bytes, uint64_t* output_bytes,
                                  uint64_t* lookup_bytes);

/* Execution knobs of the colour stage; none of them changes a result.
 *   FGPU_TUNE_ORDER_MIN_READS  passes of at least this many reads are processed in locality order (reads sorted by the rarest
 *                              colour set among their ids, so that the lists of neighbouring reads are found in the L2);
 *                              UINT64_MAX = never, the default: measured in round 3, the order cuts the fetched bytes of the
 *                              intersection kernel by 3.8x and makes nothing faster. Environment: FULGOR_ORDER=1
 *                              (then passes of at least FULGOR_ORDER_MIN_READS = 16384 reads are ordered).
 *   FGPU_TUNE_SMALL_RESULTS    1 (default; environment FULGOR_SMALL=0): full-intersection results of at most 16 colours travel
 *                              between the intersection and the expansion kernel as colours instead of as a bitmap row.
 *   FGPU_TUNE_DENSE_ROWS       1 (default; environment FULGOR_DENSE_ROWS=0): the full intersection of a hybrid index runs on
 *                              dense rows (every colour set as a plain bitmap row in HBM, built at load while they fit a quarter
 *                              of the device's memory: FULGOR_ROWS_MAX_BYTES); 0: on the packed blocks of the gap-coded lists.
 *   FGPU_TUNE_DEDUPLICATE      0 (default; environment FULGOR_DEDUPLICATE=1): 1 = `--deduplicate` (tools/pseudoalign.cpp:91-226): the full
 *                              intersection of a pass runs once per DISTINCT list of colour-set ids — the reads are ordered by a hash
 *                              of their id lists on the device, neighbours compared exactly, every read takes its group's result.
 *                              Same results, in file order, in every format.
 */
enum { FGPU_TUNE_ORDER_MIN_READS = 0, FGPU_TUNE_SMALL_RESULTS = 1, FGPU_TUNE_DENSE_ROWS = 2, FGPU_TUNE_DEDUPLICATE = 3 };
/* distinct id lists of the last pass of `res` under FGPU_TUNE_DEDUPLICATE (0: the pass was not deduplicated) */
int fgpu_result_distinct_lists(const fgpu_result* res, uint64_t* num_lists);
int fgpu_tune(fgpu_index* idx, int knob, uint64_t value);

/* One line about the device a handle lives on — ordinal, name, PCI address, CUs, free / total memory, the host NUMA node, and which
 * copy engines the library chose for the bulk copies of the streamed path with the rates it measured (csrc/copy_engines.hip.h) —
 * what every rank of a multi-GPU run prints when it starts. malloc'd: fgpu_free. */
int fgpu_device_report(fgpu_index* idx, char** out);
/* the rule behind that choice, exposed for tests: of n engines with the rates measured, those that carry host traffic at full rate
 * (within half of the best) — three or four of them, or none when the measurement is ambiguous (then the default engines are used).
 * fast: room for n entries. */
int fgpu_copy_engines_classify(const uint32_t* engines, const double* gb_per_s, uint32_t n, uint32_t* fast, uint32_t* num_fast);

/* per-kernel HIP-event timing on the engine's stream (FGPU_K_H2D / FGPU_K_D2H: the copies of the streaming worker loop and of
 * the device-side formatters, bracketed the same way) */
enum { FGPU_K_LOOKUP = 0, FGPU_K_INTERSECT = 1, FGPU_K_UNION = 2, FGPU_K_SCAN = 3, FGPU_K_EXPAND = 4,
       FGPU_K_HITS = 5, FGPU_K_DESC = 6, FGPU_K_FORMAT = 7, FGPU_K_ORDER = 8, FGPU_K_H2D = 9, FGPU_K_D2H = 10, FGPU_K_COUNT = 11 };
int fgpu_timing_enable(fgpu_index* idx, int on);
int fgpu_timing_reset(fgpu_index* idx);
int fgpu_timing_get(fgpu_index* idx, int kernel, double* total_ms, uint64_t* launches);
const char* fgpu_kernel_name(int kernel);

/* ---- output formatters (src/ps_utils.cpp:48-243): host side, one call per batch of CSR results ----------
 * ascii "<id>\t<count>[\t<colour>...]\n"; binary u32 id, u32 count, u32 x count; compressed = the
 * reference's bit-packed blocks (u64 num_colors header, then {u64 num_bits, words} blocks closed past 2^14
 * bytes). Buffers are malloc'd: release with fgpu_free. fgpu_formatter_finish flushes and destroys. */
typedef struct fgpu_formatter fgpu_formatter;
enum { FGPU_FMT_ASCII = 0, FGPU_FMT_BINARY = 1, FGPU_FMT_COMPRESSED = 2 };
int fgpu_formatter_create(int format, uint64_t num_colors, fgpu_formatter** out, char** header, uint64_t* header_len);
int fgpu_formatter_add(fgpu_formatter* f, uint32_t first_id, const uint64_t* offsets, const uint32_t* colors, uint64_t n,
                       char** out, uint64_t* out_len);
int fgpu_formatter_finish(fgpu_formatter* f, char** out, uint64_t* out_len);

/* Query reader (src/ps_utils.cpp:245-305): FASTA / FASTQ, plain or gzip, parsed natively off the caller's thread
 * (inflate + parse overlap with the GPU passes). Plain files are mapped and parsed by `threads` threads at once, byte range by
 * byte range; a gzip stream is inflated and parsed by one thread. fgpu_fastx_next returns the next at most max_reads reads in
 * file order (read id = position in the file, as FQFeeder numbers them) as concatenated bases + (n + 1) offsets.
 * The two buffers belong to the reader (pinned host memory when a HIP device is present) and stay valid for the next THREE
 * calls on it (a ring of four batches: a worker loop keeps several passes in flight); *n = 0 at end of file.
 * A block-compressed gzip file (BGZF, as bgzip writes it: gzip members of at most 64 KB that announce their size) is inflated
 * member by member by the same threads and parsed like a plain file.
 * fgpu_fastx_open_part reads only the records that start in the byte range [begin, end) of the TEXT of a plain or
 * block-compressed file (both ends are moved forward to the next record boundary, so consecutive ranges partition the
 * file): every GPU of a multi-GPU run takes one part; fgpu_fastx_text_size gives the length of that text (the file size, or
 * the inflated size) and whether the file can be read in parts at all (an ordinary gzip stream cannot); fgpu_fastx_count
 * returns the number of records of a part (their global read ids follow from the counts of the parts in front).
 * threads = 0: half of the host's hardware threads, at most 24. */
typedef struct fgpu_fastx fgpu_fastx;
int fgpu_fastx_open(const char* path, fgpu_fastx** out);
int fgpu_fastx_open_part(const char* path, unsigned threads, uint64_t begin, uint64_t end, fgpu_fastx** out);
int fgpu_fastx_text_size(const char* path, uint64_t* size, int* can_be_read_in_parts);
int fgpu_fastx_count(const char* path, unsigned threads, uint64_t begin, uint64_t end, uint64_t* num_reads);
/* number of records of an open reader's part, by a walk over the record grammar on the reader's threads that copies nothing and
 * does not consume the reader: the ranks of a multi-GPU run open their part ONCE, count it, exchange the counts (read ids are
 * file order) and then stream it. Fails for a source that has to be read to be counted (ordinary gzip, wrapped FASTQ). */
int fgpu_fastx_count_part(fgpu_fastx* f, uint64_t* num_reads);
int fgpu_fastx_next(fgpu_fastx* f, uint64_t max_reads, const char** bases, const uint64_t** offs, uint64_t* n);
/* names of the records of the last batch (kseq's name: the header up to the first blank), concatenated + (n + 1) offsets;
 * same lifetime as the batch */
int fgpu_fastx_names(fgpu_fastx* f, const char** names, const uint64_t** name_offs);
void fgpu_fastx_close(fgpu_fastx* f);
/* batches of a reader that stay valid at a time (the ring behind fgpu_fastx_next): a worker loop may keep this many minus one
 * passes in flight */
int fgpu_fastx_ring(void);

/* pseudoalign_orchestrator + pseudoalign_worker (tools/pseudoalign.cpp:12-89) for one query file (or one part of it): every
 * record of the open reader `query` is pseudoaligned (algo / tau as in fgpu_run) and its result written to the file descriptor
 * out_fd in `format` (FGPU_FMT_*), records in file order, read ids counting from first_read_id (src/ps_utils.cpp:276,286: read id =
 * position in the file); write_header != 0 puts the compressed format's 8-byte file header in front. out_fd < 0: nothing is
 * formatted or written (counters only). Returns when everything is written. The loop keeps `workers` batches of at most
 * batch_reads reads in flight (0 = defaults: 5, and 2^18 for compressed records, 2^15 for ascii and binary ones): the reader's threads parse byte ranges of the file into pinned
 * memory, each range goes to the device as it lies, lookup -> colour stage -> device-side formatter -> copy out run per batch on
 * the batch's own stream, so that the copy in of one batch, the kernels of another and the copy out of a third overlap. The u32
 * colour lists are not built for the compressed format. num_reads / num_mapped: the two counters of ps_options
 * (src/ps_utils.cpp:417-448). The reader is consumed; it must not be used with fgpu_fastx_next at the same time. */
int fgpu_pseudoalign_stream(fgpu_index* idx, fgpu_fastx* query, int out_fd, int algo, double tau, int format, uint64_t first_read_id,
                            int write_header, uint64_t batch_reads, unsigned workers, uint64_t* num_reads, uint64_t* num_mapped);
/* Opt-in preparation of a process that is about to run fgpu_pseudoalign_stream ONCE (one `pseudoalign` command: the reference's only
 * published figure is one command, tools/pseudoalign.cpp:340-369). The first run of a process otherwise pays for pinning 0.4-0.7 GB of
 * host memory (0.16 ms per megabyte, and pinning stalls the copies in flight) and for creating the workers' streams and buffers.
 *   fgpu_prepare_host needs no index: it starts the HIP runtime on `device` and pins the host buffers the loop will use — the reader's
 *     chunks (reader_threads + 8 ranges parsed ahead and `workers` batches of batch_reads reads of text_bytes_per_read bytes of text each,
 *     FASTQ or FASTA) and, if out_bytes_per_read is given, one output buffer per worker — into the process-wide pool the reader and the
 *     results draw from. Meant to be called on a thread of its own WHILE fgpu_open runs on another.
 *   fgpu_stream_prepare creates the workers' results (streams, device buffers sized for batches of batch_reads reads of at most
 *     max_read_bases bases, output buffers if out_bytes_per_read is given) and keeps them with the index, where the loop finds them.
 * 0 for reader_threads / workers / batch_reads: the loop's defaults. Neither is ever called by the library itself: a library user
 * that never streams is not charged the pinned memory. */
int fgpu_prepare_host(int device, unsigned reader_threads, unsigned workers, uint64_t batch_reads, uint64_t text_bytes_per_read, int fastq,
                      uint64_t out_bytes_per_read, uint64_t total_text_bytes /* of the query, 0 = unknown: a small query pins less */);
int fgpu_stream_prepare(fgpu_index* idx, int format, uint64_t batch_reads, unsigned workers, uint32_t max_read_bases, uint64_t out_bytes_per_read);
/* timeline of the last fgpu_pseudoalign_stream of this process as text (per batch: when it was acquired from the parser, queued,
 * through the colour stage, formatted and copied out, written), plus what the parser threads spent; malloc'd: fgpu_free */
int fgpu_last_stream_report(char** out);

/* Device-side formatting of the last pass of `res` (src/ps_utils.cpp:48-135): the records of reads
 * first_read_id .. first_read_id + n - 1 in file order, ascii ("<id>\t<count>[\t<colour>...]\n") or binary (u32 id, u32
 * count, u32 x count), built by HIP kernels from the resident CSR and copied to a malloc'd host buffer (fgpu_free).
 * Byte-identical to fgpu_formatter_add on the downloaded CSR. FGPU_FMT_COMPRESSED (src/ps_utils.cpp:158-239) is built from the
 * result bitmaps: the same records in blocks of 256 (the reference's block cuts depend on its workers' buffers; any cut
 * is the same format); the caller writes the 8-byte file header (fgpu_formatter_create) once in front. */
int fgpu_result_format(const fgpu_result* res, int format, uint32_t first_read_id, char** out, uint64_t* out_len);
/* Same records without the extra copy: *out points into a pinned host buffer owned by `res` (the D2H copy runs at PCIe
 * speed and the buffer is recycled); valid until the next format call on `res` or fgpu_result_free. */
int fgpu_result_format_view(const fgpu_result* res, int format, uint32_t first_read_id, const char** out, uint64_t* out_len);

/* `fulgor dump` (src/index.cpp:59-120): writes <basename>.metadata.txt / .filenames.txt / .unitigs.fa / .color_sets.txt,
 * the reference's text interchange format (the same files fgpu_open reads): indexes travel in both directions. */
int fgpu_dump(const fgpu_index* idx, const char* basename);

/* ---- index export (lets tests hand the same encoded index to the oracle) -------------------------- */
int fgpu_export_sizes(const fgpu_index* idx, uint64_t* unitig_bases, uint64_t* num_unitigs, uint64_t* color_words,
                      uint64_t* color_bits, uint64_t* num_sets);
int fgpu_export(const fgpu_index* idx, char* unitig_bases, uint64_t* unitig_off, uint32_t* unitig_csid,
                uint64_t* color_words, uint64_t* color_offsets, uint32_t* thresholds /* n, sparse, dense */);

#ifdef __cplusplus
}
#endif
#endif
