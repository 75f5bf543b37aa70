// Package spaghetti is the cgo binding of libspaghetti_rank.so (include/spaghetti_rank.h).
//
// It is the ONLY file that touches C.  go/ranking and go/retrieval keep the reference's
// exported signatures (ranking/pagerank.go:14, ranking/term_weighting.go:10,
// retrieval/main_retrieve.go:15) and call into this package.
//
// NOTE: written against the header but NOT compiled in the build image (no Go toolchain
// there, SURVEY.md §8c); the same entry points are exercised through Python ctypes
// (spaghettisearch_amd/_lib.py) by the test-suite.
//
// Error policy: every non-zero status becomes panic(err), like the reference
// (pagerank.go:20,29,49,57,76,81; term_weighting.go:14,23,34,48,52).
package spaghetti

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../spaghettisearch_amd -lspaghetti_rank -Wl,-rpath,${SRCDIR}/../../spaghettisearch_amd
#include <stdlib.h>
#include "spaghetti_rank.h"
*/
import "C"

import (
	"fmt"
	"math"
	"os"
	"strconv"
	"sync"
	"time"
	"unsafe"
)

// Per-request limits of the scorer (a batch that breaks one is refused as a whole: callers validate first).
const (
	MaxQueryTerms  = int(C.SS_MAX_QUERY_TERMS)
	MaxPhraseTerms = int(C.SS_MAX_PHRASE_TERMS)
)

// Hit mirrors ss_hit.
type Hit struct {
	Doc      uint32
	Title    float64
	Body     float64
	PageRank float64
	Final    float64
}

type Ctx struct{ h *C.ss_ctx }
type Graph struct {
	h   *C.ss_graph
	ctx *Ctx
	N   uint64
}
type Index struct {
	h      *C.ss_index
	ctx    *Ctx
	NDocs  uint64
	NTerms uint64
	NPost  uint64
}
type Scorer struct {
	h   *C.ss_scorer
	ctx *Ctx
}

var (
	once   sync.Once
	global *Ctx
)

// statusErr turns a library status into an error (nil for SS_OK).
func statusErr(c *Ctx, rc C.int32_t, what string) error {
	if rc == C.SS_OK {
		return nil
	}
	var h *C.ss_ctx
	if c != nil {
		h = c.h
	}
	return fmt.Errorf("%s: status %d: %s", what, int(rc), C.GoString(C.ss_last_error(h)))
}

func check(c *Ctx, rc C.int32_t, what string) {
	if err := statusErr(c, rc, what); err != nil {
		panic(err)
	}
}

// JobInfo is this process's place in a multi-GPU job (SS_RANK / SS_WORLD; absent = a job of one).
type JobInfo struct{ Rank, World int }

func Job() JobInfo {
	w, _ := strconv.Atoi(os.Getenv("SS_WORLD"))
	r, _ := strconv.Atoi(os.Getenv("SS_RANK"))
	if w < 2 {
		return JobInfo{0, 1}
	}
	return JobInfo{r, w}
}

// Default returns the process-wide context (one process per GPU; the launcher pins the GPU with HIP_VISIBLE_DEVICES, so the
// device index is 0).  In a multi-GPU job it also joins the job's RCCL communicator through SS_COMM_ID_FILE:
//   - the file holds the 128-byte id followed by the job's nonce SS_JOB_ID (the launcher gives every start of the job a
//     fresh value, e.g. its pid + start time); a reader accepts the file only when the nonce is its own, so a file left
//     behind by an earlier or crashed job is never used;
//   - rank 0 removes whatever is at the path first, writes temp file + rename, and removes the file again once
//     CommInit has returned (ncclCommInitRank is collective: every rank holds the id by then);
//   - the other ranks wait SS_COMM_TIMEOUT_S seconds at most (default 120) and then panic, so that a job whose rank 0
//     never came up exits non-zero instead of hanging.
func Default() *Ctx {
	once.Do(func() {
		// the header this package was compiled against and the library that got loaded must be the same ABI (4: submit / collect,
		// late-completing ss_graph_create)
		if v := int(C.ss_abi_version()); v != int(C.SS_ABI_VERSION) {
			panic(fmt.Errorf("libspaghetti_rank: ABI version %d, this package was built against %d", v, int(C.SS_ABI_VERSION)))
		}
		var h *C.ss_ctx
		check(nil, C.ss_init(0, &h), "ss_init")
		global = &Ctx{h}
		if job := Job(); job.World > 1 {
			path := os.Getenv("SS_COMM_ID_FILE")
			if path == "" {
				panic(fmt.Errorf("SS_WORLD=%d needs SS_COMM_ID_FILE", job.World))
			}
			nonce := []byte(os.Getenv("SS_JOB_ID"))
			if len(nonce) == 0 {
				panic(fmt.Errorf("SS_WORLD=%d needs SS_JOB_ID (a value unique to this start of the job)", job.World))
			}
			var id []byte
			if job.Rank == 0 {
				_ = os.Remove(path)
				id = CommUniqueID()
				if err := os.WriteFile(path+".tmp", append(append([]byte{}, id...), nonce...), 0o600); err != nil {
					panic(err)
				}
				if err := os.Rename(path+".tmp", path); err != nil {
					panic(err)
				}
			} else {
				limit := 120
				if v, err := strconv.Atoi(os.Getenv("SS_COMM_TIMEOUT_S")); err == nil && v > 0 {
					limit = v
				}
				deadline := time.Now().Add(time.Duration(limit) * time.Second)
				for id == nil {
					b, err := os.ReadFile(path)
					if err == nil && len(b) == C.SS_COMM_ID_BYTES+len(nonce) && string(b[C.SS_COMM_ID_BYTES:]) == string(nonce) {
						id = b[:C.SS_COMM_ID_BYTES]
						break
					}
					if time.Now().After(deadline) {
						panic(fmt.Errorf("rank %d: no communicator id for job %q in %s after %d s", job.Rank, string(nonce), path, limit))
					}
					time.Sleep(50 * time.Millisecond)
				}
			}
			global.CommInit(id, job.Rank, job.World)
			if job.Rank == 0 {
				_ = os.Remove(path)
			}
		}
	})
	return global
}

// OptionDefault restores an option's default (SS_OPTION_DEFAULT).
const OptionDefault = int64(math.MinInt64)

// SetOption sets a named tuning option of the context (ss_set_option; the names are listed in the header).
func (c *Ctx) SetOption(name string, value int64) {
	cs := C.CString(name)
	defer C.free(unsafe.Pointer(cs))
	check(c, C.ss_set_option(c.h, cs, C.int64_t(value)), "ss_set_option")
}

// CommSplit replaces the context's communicator by the sub-communicator of the ranks that pass the same color (2-D
// decomposition: topic groups x doc shards).
func (c *Ctx) CommSplit(color, key int) {
	check(c, C.ss_comm_split(c.h, C.int32_t(color), C.int32_t(key)), "ss_comm_split")
}

func u64p(s []uint64) *C.uint64_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint64_t)(unsafe.Pointer(&s[0]))
}
func u32p(s []uint32) *C.uint32_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&s[0]))
}
func i32p(s []int32) *C.int32_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.int32_t)(unsafe.Pointer(&s[0]))
}
func f32p(s []float32) *C.float {
	if len(s) == 0 {
		return nil
	}
	return (*C.float)(unsafe.Pointer(&s[0]))
}
func f64p(s []float64) *C.double {
	if len(s) == 0 {
		return nil
	}
	return (*C.double)(unsafe.Pointer(&s[0]))
}

// NewGraph uploads the out-edge CSR of forw[2] (parent -> children).  The library copies the
// slices before returning (cgo pointer rules), so they may be garbage collected afterwards.
func (c *Ctx) NewGraph(outPtr []uint64, outDst []uint32) *Graph {
	n := uint64(len(outPtr) - 1)
	var h *C.ss_graph
	check(c, C.ss_graph_create(c.h, C.uint64_t(n), C.uint64_t(len(outDst)), u64p(outPtr), u32p(outDst), 0, 1, &h), "ss_graph_create")
	return &Graph{h, c, n}
}
func (g *Graph) Close() { C.ss_graph_destroy(g.h) }

// ApplyDelta patches the resident link graph: the out-edges of changed[i] become newChildren[newPtr[i]:newPtr[i+1]],
// nNodesNew >= N admits new pages (ss_graph_apply_delta; indexer.go:301-304 rewrites forw[2] of a re-crawled page).
func (g *Graph) ApplyDelta(nNodesNew uint64, changed []uint32, newPtr []uint64, newChildren []uint32) {
	check(g.ctx, C.ss_graph_apply_delta(g.h, C.uint64_t(nNodesNew), C.uint64_t(len(changed)), u32p(changed), u64p(newPtr), u32p(newChildren)),
		"ss_graph_apply_delta")
	g.N = nNodesNew
}

// PageRankTeleport is the opt-in topic-sensitive run (SURVEY.md §8f-3): sets[k] = DISTINCT node ids of topic k's
// teleport set (empty = the reference's uniform teleport for that topic).  rank[k*N+v], iters[k].
func (g *Graph) PageRankTeleport(d, eps float64, nTopic []int32, sets [][]uint32) ([]float64, []int32) {
	k := len(nTopic)
	var pr *C.ss_pr
	check(g.ctx, C.ss_pr_create(g.h, C.double(d), C.double(eps), 0, C.int32_t(k), i32p(nTopic), &pr), "ss_pr_create")
	defer C.ss_pr_destroy(pr)
	setPtr := make([]uint64, k+1)
	var nodes []uint32
	for i, s := range sets {
		nodes = append(nodes, s...)
		setPtr[i+1] = uint64(len(nodes))
	}
	check(g.ctx, C.ss_pr_set_teleport(pr, u64p(setPtr), u32p(nodes)), "ss_pr_set_teleport")
	check(g.ctx, C.ss_pr_begin(pr), "ss_pr_begin")
	iters := make([]int32, k)
	var nActive, sweeps C.int32_t = C.int32_t(k), 0
	for nActive > 0 {
		check(g.ctx, C.ss_pr_step(pr, 8), "ss_pr_step")
		check(g.ctx, C.ss_pr_status(pr, i32p(iters), &nActive, &sweeps, nil, nil), "ss_pr_status")
	}
	rank := make([]float64, uint64(k)*g.N)
	check(g.ctx, C.ss_pr_read(pr, f64p(rank)), "ss_pr_read")
	return rank, iters
}

// ---- several GPUs: one process (= one context) per GPU, collectives inside the library (RCCL over xGMI) ----
//
// SS_RANK / SS_WORLD / SS_COMM_ID_FILE in the environment describe the job (a launcher starts one crawl process per
// GPU with SS_RANK = 0..SS_WORLD-1 and HIP_VISIBLE_DEVICES set to its GPU).  Rank 0 writes the 128-byte communicator
// id to SS_COMM_ID_FILE, the others wait for the file.  Without SS_WORLD the process is the whole job (world 1).

// CommInit joins this context to the job's communicator.  Blocks until every rank has joined.
func (c *Ctx) CommInit(id []byte, rank, world int) {
	if len(id) != C.SS_COMM_ID_BYTES {
		panic(fmt.Errorf("communicator id must be %d bytes", int(C.SS_COMM_ID_BYTES)))
	}
	check(c, C.ss_comm_init(c.h, unsafe.Pointer(&id[0]), C.int32_t(rank), C.int32_t(world)), "ss_comm_init")
}

// CommUniqueID is called by rank 0 only.
func CommUniqueID() []byte {
	id := make([]byte, C.SS_COMM_ID_BYTES)
	check(nil, C.ss_comm_unique_id(unsafe.Pointer(&id[0])), "ss_comm_unique_id")
	return id
}

// AllReduceU64 sums buf over the ranks in place (whole-corpus document frequencies of a doc-range-sharded index).
func (c *Ctx) AllReduceU64(buf []uint64) {
	check(c, C.ss_comm_allreduce_u64(c.h, u64p(buf), C.uint64_t(len(buf))), "ss_comm_allreduce_u64")
}

// NewGraphShard uploads the WHOLE out-edge CSR and keeps the destination rows of shard `rank` of `world`.
func (c *Ctx) NewGraphShard(outPtr []uint64, outDst []uint32, rank, world int) *Graph {
	n := uint64(len(outPtr) - 1)
	var h *C.ss_graph
	check(c, C.ss_graph_create(c.h, C.uint64_t(n), C.uint64_t(len(outDst)), u64p(outPtr), u32p(outDst), C.int32_t(rank), C.int32_t(world), &h),
		"ss_graph_create")
	return &Graph{h, c, n}
}

// PageRankSharded runs this rank's part of the doc-range-sharded power iteration (one RCCL all-gather per sweep inside
// the library): ids[r] = node id of local row r, rank[k*rows+r], iters[k].  Every process writes its own rows of forw[3].
func (g *Graph) PageRankSharded(d, eps float64, nTopic []int32) ([]uint32, []float64, []int32) {
	var info C.ss_graph_info
	check(g.ctx, C.ss_graph_get_info(g.h, &info), "ss_graph_get_info")
	rows := uint64(info.n_rows_local)
	k := len(nTopic)
	ids := make([]uint32, rows)
	rank := make([]float64, uint64(k)*rows)
	iters := make([]int32, k)
	check(g.ctx, C.ss_pagerank_run_sharded(g.h, C.double(d), C.double(eps), 0, C.int32_t(k), i32p(nTopic), 0, u32p(ids), f64p(rank), i32p(iters)),
		"ss_pagerank_run_sharded")
	return ids, rank, iters
}

// PageRank runs all topics to convergence: rank[k*N+v], iters[k].
func (g *Graph) PageRank(d, eps float64, nTopic []int32) ([]float64, []int32) {
	k := len(nTopic)
	rank := make([]float64, uint64(k)*g.N)
	iters := make([]int32, k)
	check(g.ctx, C.ss_pagerank_run(g.h, C.double(d), C.double(eps), 0, C.int32_t(k), i32p(nTopic), f64p(rank), i32p(iters)), "ss_pagerank_run")
	return rank, iters
}

func (c *Ctx) NewIndex(nDocs uint64, termPtr []uint64, postDoc []uint32, postTf []float32) *Index {
	var h *C.ss_index
	nT := uint64(len(termPtr) - 1)
	check(c, C.ss_index_create(c.h, C.uint64_t(nDocs), C.uint64_t(nT), u64p(termPtr), u32p(postDoc), f32p(postTf), &h), "ss_index_create")
	return &Index{h, c, nDocs, nT, uint64(len(postDoc))}
}
func (ix *Index) Close() { check(ix.ctx, C.ss_index_destroy(ix.h), "ss_index_destroy") }

// TfIdfBuild = ranking.UpdateTermWeights' arithmetic; returns the new weights and magnitudes.
func (ix *Index) TfIdfBuild(totalDocs uint64) (w []float32, mag []float64) {
	w = make([]float32, ix.NPost)
	mag = make([]float64, ix.NDocs)
	check(ix.ctx, C.ss_tfidf_build(ix.h, C.uint64_t(totalDocs), f32p(w), f64p(mag), nil), "ss_tfidf_build")
	return
}
func (ix *Index) SetPositions(posPtr []uint64, pos []float32) {
	check(ix.ctx, C.ss_index_set_positions(ix.h, u64p(posPtr), f32p(pos)), "ss_index_set_positions")
}

// SetDocFreq: multi-GPU doc-range shards only — df[t] = length of term t's whole posting list
// (term_weighting.go:37 len(docs)), summed over the shards by the caller.  Call before TfIdfBuild.
func (ix *Index) SetDocFreq(df []uint64) {
	check(ix.ctx, C.ss_index_set_doc_freq(ix.h, u64p(df)), "ss_index_set_doc_freq")
}

// Delta is what indexer.checkAndUpdate (indexer/indexer.go:420-641) and the re-index after it change in ONE inverted
// table: DelDocs lose every posting (the changed page's old words, :455-531), the (DelTerm[i], DelDoc[i]) postings go
// (anchor words of its children, :533-616), the (AddTerm[i], AddDoc[i], AddW[i]) postings arrive.
type Delta struct {
	DelDocs         []uint32
	DelTerm, DelDoc []uint32
	AddTerm, AddDoc []uint32
	AddW            []float32
	AddPosPtr       []uint64  // optional: positions of the new postings, AddPosPtr[len(AddTerm)+1] into AddPos
	AddPos          []float32 // listPos[1:] (parser.go:195-207)
}

// ApplyDelta merges d into the resident table on the device (no BadgerDB row rewrite, no re-upload).  Scorers on this
// table must be closed before and created again after; call RefreshMagnitudes or TfIdfBuild next, as
// start_crawl.go:176-177 re-runs UpdateTermWeights after every crawl.
//
// When the table's squared magnitudes are resident (after TfIdfBuild or RefreshMagnitudes) the delta keeps the magnitudes
// of the docs it touches up to date itself; ReadMagnitudes returns them for the forw[4] rows.  New words / new child
// pages: Resize first.
func (ix *Index) ApplyDelta(d Delta) {
	check(ix.ctx, C.ss_index_apply_delta_pos(ix.h, C.uint64_t(len(d.DelDocs)), u32p(d.DelDocs),
		C.uint64_t(len(d.DelTerm)), u32p(d.DelTerm), u32p(d.DelDoc),
		C.uint64_t(len(d.AddTerm)), u32p(d.AddTerm), u32p(d.AddDoc), f32p(d.AddW), u64p(d.AddPosPtr), f32p(d.AddPos)), "ss_index_apply_delta_pos")
	var nPost C.uint64_t
	check(ix.ctx, C.ss_index_get_info(ix.h, nil, nil, &nPost), "ss_index_get_info")
	ix.NPost = uint64(nPost)
}

// Resize grows the doc and / or term space of the resident table (ss_index_resize).
func (ix *Index) Resize(nDocs, nTerms uint64) {
	check(ix.ctx, C.ss_index_resize(ix.h, C.uint64_t(nDocs), C.uint64_t(nTerms)), "ss_index_resize")
	ix.NDocs, ix.NTerms = nDocs, nTerms
}

// ReadMagnitudes returns the magnitudes of docs as they stand on the device.
func (ix *Index) ReadMagnitudes(docs []uint32) []float64 {
	mag := make([]float64, len(docs))
	check(ix.ctx, C.ss_index_read_magnitudes(ix.h, C.uint64_t(len(docs)), u32p(docs), f64p(mag)), "ss_index_read_magnitudes")
	return mag
}
func (ix *Index) RefreshMagnitudes() []float64 {
	mag := make([]float64, ix.NDocs)
	check(ix.ctx, C.ss_index_refresh_magnitudes(ix.h, f64p(mag)), "ss_index_refresh_magnitudes")
	return mag
}
func (ix *Index) SetWeighted(mag []float64) {
	check(ix.ctx, C.ss_index_set_weighted(ix.h, f64p(mag)), "ss_index_set_weighted")
}

func (c *Ctx) NewScorer(title, body *Index) *Scorer {
	var h *C.ss_scorer
	check(c, C.ss_scorer_create(c.h, title.h, body.h, &h), "ss_scorer_create")
	return &Scorer{h, c}
}
func (s *Scorer) Close() { C.ss_scorer_destroy(s.h) }
func (s *Scorer) SetPrior(kTopics int, rank []float64) {
	check(s.ctx, C.ss_scorer_set_prior(s.h, C.int32_t(kTopics), f64p(rank)), "ss_scorer_set_prior")
}

// ScoreTopKPhrase = ScoreTopK plus one (concatenated) quoted phrase per query (retrieval/phrase.go).
func (s *Scorer) ScoreTopKPhrase(qPtr, qTerms, pPtr, pTerms []uint32, queryLen []int32, topicProbs []float64, k int) ([][]Hit, error) {
	nq := len(qPtr) - 1
	if nq == 0 {
		return nil, nil
	}
	raw := make([]C.ss_hit, nq*k)
	nHits := make([]int32, nq)
	rc := C.ss_score_topk_phrase(s.h, C.int32_t(nq), u32p(qPtr), u32p(qTerms), u32p(pPtr), u32p(pTerms), i32p(queryLen),
		f64p(topicProbs), C.int32_t(k), (*C.ss_hit)(unsafe.Pointer(&raw[0])), i32p(nHits))
	if err := statusErr(s.ctx, rc, "ss_score_topk_phrase"); err != nil {
		return nil, err // the caller decides: a serving path must not die of one bad request
	}
	out := make([][]Hit, nq)
	for q := 0; q < nq; q++ {
		out[q] = make([]Hit, nHits[q])
		for i := range out[q] {
			r := raw[q*k+i]
			out[q][i] = Hit{uint32(r.doc), float64(r.title), float64(r.body), float64(r.pagerank), float64(r.final)}
		}
	}
	return out, nil
}

// ScoreTopK scores a batch of OR queries; safe to call from many goroutines (the library
// serialises calls on one context).  topicProbs may be nil (reference default: sqd = 0).
func (s *Scorer) ScoreTopK(qPtr, qTerms []uint32, queryLen []int32, topicProbs []float64, k int) ([][]Hit, error) {
	nq := len(qPtr) - 1
	raw := make([]C.ss_hit, nq*k)
	nHits := make([]int32, nq)
	rc := C.ss_score_topk(s.h, C.int32_t(nq), u32p(qPtr), u32p(qTerms), i32p(queryLen), f64p(topicProbs), C.int32_t(k),
		(*C.ss_hit)(unsafe.Pointer(&raw[0])), i32p(nHits))
	if err := statusErr(s.ctx, rc, "ss_score_topk"); err != nil {
		return nil, err
	}
	out := make([][]Hit, nq)
	for q := 0; q < nq; q++ {
		out[q] = make([]Hit, nHits[q])
		for i := range out[q] {
			r := raw[q*k+i]
			out[q][i] = Hit{uint32(r.doc), float64(r.title), float64(r.body), float64(r.pagerank), float64(r.final)}
		}
	}
	return out, nil
}

// Ticket names a batch in flight (Submit .. Collect).
type Ticket struct {
	id    C.uint64_t
	nq, k int
}

// ScoreInflight is the number of submitted batches one scorer holds at a time (SS_SCORE_INFLIGHT): a further Submit is refused
// with SS_ERR_STATE until one has been collected.
const ScoreInflight = int(C.SS_SCORE_INFLIGHT)

// Submit enqueues a batch of queries (pPtr / pTerms nil: no quoted phrases) and returns at once; up to C.SS_SCORE_INFLIGHT batches may be in flight.  A server
// goroutine that has the next batch of requests ready calls Submit for it before it Collects the previous one: the host-side
// plan of batch i+1 and the copy-out of batch i-1 then run under the kernels of batch i.
func (s *Scorer) Submit(qPtr, qTerms, pPtr, pTerms []uint32, queryLen []int32, topicProbs []float64, k int) (Ticket, error) {
	nq := len(qPtr) - 1
	var id C.uint64_t
	rc := C.ss_score_topk_submit(s.h, C.int32_t(nq), u32p(qPtr), u32p(qTerms), u32p(pPtr), u32p(pTerms), i32p(queryLen), f64p(topicProbs), C.int32_t(k), &id)
	if err := statusErr(s.ctx, rc, "ss_score_topk_submit"); err != nil {
		return Ticket{}, err
	}
	return Ticket{id, nq, k}, nil
}

// Collect waits for the batch of t and returns its hits like ScoreTopK.
func (s *Scorer) Collect(t Ticket) ([][]Hit, error) {
	raw := make([]C.ss_hit, t.nq*t.k+1)
	nHits := make([]int32, t.nq+1)
	rc := C.ss_score_topk_collect(s.h, t.id, (*C.ss_hit)(unsafe.Pointer(&raw[0])), i32p(nHits))
	if err := statusErr(s.ctx, rc, "ss_score_topk_collect"); err != nil {
		return nil, err
	}
	out := make([][]Hit, t.nq)
	for q := 0; q < t.nq; q++ {
		out[q] = make([]Hit, nHits[q])
		for i := range out[q] {
			r := raw[q*t.k+i]
			out[q][i] = Hit{uint32(r.doc), float64(r.title), float64(r.body), float64(r.pagerank), float64(r.final)}
		}
	}
	return out, nil
}

// MergeHits: multi-GPU doc-range shards only — parts[p] holds shard p's rows of one query batch as returned
// by ScoreTopK (local doc ids), docBase[p] the shard's first corpus doc id.  Returns the k best of the union
// per query in the order of appendSort (util.go:48-54).
func (c *Ctx) MergeHits(parts [][][]Hit, docBase []uint32, k int) [][]Hit {
	np := len(parts)
	nq := len(parts[0])
	raw := make([]C.ss_hit, np*nq*k)
	nHits := make([]int32, np*nq)
	for p := range parts {
		for q := range parts[p] {
			nHits[p*nq+q] = int32(len(parts[p][q]))
			for i, h := range parts[p][q] {
				r := &raw[(p*nq+q)*k+i]
				r.doc, r.title, r.body, r.pagerank, r.final = C.uint32_t(h.Doc), C.double(h.Title), C.double(h.Body), C.double(h.PageRank), C.double(h.Final)
			}
		}
	}
	out := make([]C.ss_hit, nq*k)
	nOut := make([]int32, nq)
	check(c, C.ss_merge_hits(c.h, C.int32_t(nq), C.int32_t(np), C.int32_t(k), (*C.ss_hit)(unsafe.Pointer(&raw[0])), i32p(nHits),
		u32p(docBase), (*C.ss_hit)(unsafe.Pointer(&out[0])), i32p(nOut)), "ss_merge_hits")
	res := make([][]Hit, nq)
	for q := 0; q < nq; q++ {
		res[q] = make([]Hit, nOut[q])
		for i := range res[q] {
			r := out[q*k+i]
			res[q][i] = Hit{uint32(r.doc), float64(r.title), float64(r.body), float64(r.pagerank), float64(r.final)}
		}
	}
	return res
}
