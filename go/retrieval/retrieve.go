// Package retrieval — drop-in for retrieval.Retrieve (retrieval/main_retrieve.go:15).
//
// The query-time index lives on the GPU: on first use the tables are flattened once
// (inv[0]/inv[1] postings, forw[4] magnitudes, forw[3] ranks) and uploaded; every Retrieve then
// costs string parsing on the host plus ONE library call, and only the k winners are decorated
// (the reference builds DocInfo + summary for every candidate, get_metadata.go:21-28).
// Rank_combined (util.go:25-36), getPhrase (util.go:151-160), getDocInfo/getSummary
// (get_metadata.go:79-235) and parser.Laundry stay as in the reference and are not repeated here.
package retrieval

import (
	"context"
	"crypto/md5"
	"encoding/hex"
	"encoding/json"
	"fmt"
	"sort"
	"strings"
	"sync"
	"time"

	db "github.com/nwihardjo/SpaghettiSearch/database"
	"github.com/nwihardjo/SpaghettiSearch/parser"

	"github.com/nwihardjo/SpaghettiSearch/go/spaghetti"
)

const topK = 50 // main_retrieve.go:99-100

// deviceIndex is ONE immutable snapshot of the tables on the device.  Requests are tokenised against a snapshot, carry it
// through the batcher and map the winners back through the same snapshot; Refresh swaps in nothing (the next Retrieve
// loads a new snapshot) and the old one is destroyed when its last user lets go.
type deviceIndex struct {
	scorer      *spaghetti.Scorer
	title, body *spaghetti.Index
	termID      map[string]uint32 // md5-hex(word) -> dense term id
	docName     []string          // dense doc id -> md5-hex(url)
	users       sync.WaitGroup    // requests in flight on this snapshot
}

func (d *deviceIndex) close() {
	d.users.Wait() // drain: no batch may still hold the scorer
	d.scorer.Close()
	d.title.Close()
	d.body.Close()
}

var (
	devMu sync.RWMutex // guards dev; held for reading only while a request registers itself on the snapshot
	dev   *deviceIndex
)

func flatten(ctx context.Context, inv db.DB, termID map[string]uint32, docID map[string]uint32) (ptr []uint64, doc []uint32, w []float32, posPtr []uint64, pos []float32) {
	comp, err := inv.Iterate(ctx)
	if err != nil {
		panic(err)
	}
	rows := make([]map[string][]float32, len(termID))
	for i := range comp.KV {
		var r map[string][]float32
		if err = json.Unmarshal(comp.KV[i].Value, &r); err != nil {
			panic(err)
		}
		rows[termID[string(comp.KV[i].Key)]] = r
	}
	posPtr = []uint64{0}
	ptr = make([]uint64, len(rows)+1)
	for i, r := range rows {
		ptr[i+1] = ptr[i] + uint64(len(r))
	}
	doc = make([]uint32, ptr[len(rows)])
	w = make([]float32, ptr[len(rows)])
	for i, r := range rows {
		type pw struct {
			d   uint32
			w   float32
			pos []float32
		}
		tmp := make([]pw, 0, len(r))
		for h, listPos := range r {
			tmp = append(tmp, pw{docID[h], listPos[0], listPos[1:]}) // [norm_tf*idf, positions...] (main_retrieve.go:227, phrase.go:144)
		}
		sort.Slice(tmp, func(a, b int) bool { return tmp[a].d < tmp[b].d })
		for j, e := range tmp {
			doc[ptr[i]+uint64(j)], w[ptr[i]+uint64(j)] = e.d, e.w
			pos = append(pos, e.pos...)
			posPtr = append(posPtr, uint64(len(pos)))
		}
	}
	return
}

func load(ctx context.Context, forw []db.DB, inv []db.DB) *deviceIndex {
	// dense ids: docs = keys of forw[3] (every PageRank node), terms = keys of inv[0] U inv[1]
	ranks, err := forw[3].Iterate(ctx)
	if err != nil {
		panic(err)
	}
	d := &deviceIndex{termID: map[string]uint32{}}
	docID := make(map[string]uint32, len(ranks.KV))
	for _, kv := range ranks.KV {
		docID[string(kv.Key)] = uint32(len(d.docName))
		d.docName = append(d.docName, string(kv.Key))
	}
	for t := 0; t < 2; t++ {
		comp, err := inv[t].Iterate(ctx)
		if err != nil {
			panic(err)
		}
		for _, kv := range comp.KV {
			if _, ok := d.termID[string(kv.Key)]; !ok {
				d.termID[string(kv.Key)] = uint32(len(d.termID))
			}
		}
	}
	n := uint64(len(d.docName))
	c := spaghetti.Default()
	tPtr, tDoc, tW, tPosPtr, tPos := flatten(ctx, inv[0], d.termID, docID)
	bPtr, bDoc, bW, bPosPtr, bPos := flatten(ctx, inv[1], d.termID, docID)
	title := c.NewIndex(n, tPtr, tDoc, tW)
	body := c.NewIndex(n, bPtr, bDoc, bW)
	title.SetPositions(tPosPtr, tPos)
	body.SetPositions(bPosPtr, bPos)
	// forw[4]: a missing "title"/"body" key reads as 0 (get_metadata.go:57-58, Q8)
	magT, magB := make([]float64, n), make([]float64, n)
	mags, err := forw[4].Iterate(ctx)
	if err != nil {
		panic(err)
	}
	for _, kv := range mags.KV {
		var m map[string]float64
		if err = json.Unmarshal(kv.Value, &m); err != nil {
			panic(err)
		}
		if id, ok := docID[string(kv.Key)]; ok {
			magT[id], magB[id] = m["title"], m["body"]
		}
	}
	title.SetWeighted(magT)
	body.SetWeighted(magB)
	d.title, d.body = title, body
	d.scorer = c.NewScorer(title, body)
	return d
}

// ---- request batching (INTEGRATION.md §4) ----------------------------------------------------------------------
// net/http runs one goroutine per request (cmd/server/server.go:47) and each calls Retrieve.  Concurrent callers are
// collected for at most batchWindow (or until maxBatch are waiting) and answered by ONE library call; every caller gets
// its own rows back and decorates its own winners.  A lone request pays the window once (1 ms against ~0.15 ms of device
// time); under load the device sees batches and runs at its batch throughput.
const (
	batchWindow = time.Millisecond
	maxBatch    = 1024
)

type reply struct {
	hits []spaghetti.Hit
	err  error
}

type request struct {
	snap           *deviceIndex
	qTerms, pTerms []uint32
	qLen           int32
	reply          chan reply // buffered: the batcher never blocks on a caller
}

var (
	reqs     = make(chan *request, 4*maxBatch)
	batchers sync.Once
)

// concat lays the batch's token lists end to end (the library's CSR form of a query batch).
func concat(batch []*request) (qPtr, qTerms, pPtr, pTerms []uint32, qLen []int32) {
	qPtr, pPtr = []uint32{0}, []uint32{0}
	qLen = make([]int32, 0, len(batch))
	for _, r := range batch {
		qTerms = append(qTerms, r.qTerms...)
		pTerms = append(pTerms, r.pTerms...)
		qPtr = append(qPtr, uint32(len(qTerms)))
		pPtr = append(pPtr, uint32(len(pTerms)))
		qLen = append(qLen, r.qLen)
	}
	return
}

// score answers `batch` (all on one snapshot) with one synchronous library call.
func score(snap *deviceIndex, batch []*request) ([][]spaghetti.Hit, error) {
	qPtr, qTerms, pPtr, pTerms, qLen := concat(batch)
	// topicProbs stays nil as in the shipped reference (main_retrieve.go:40,87-88): sqd = 0.
	return snap.scorer.ScoreTopKPhrase(qPtr, qTerms, pPtr, pTerms, qLen, nil, topK)
}

// Batches in flight: batchLoop submits a batch (the library enqueues it and returns) and goes back to collecting requests;
// collectLoop waits for the batches in the order they were submitted and answers their callers.  A batch holds one of
// spaghetti.ScoreInflight (= SS_SCORE_INFLIGHT = 3) slots from BEFORE its Submit until its Collect has returned, so that the
// library never sees a fourth ticket: under sustained load batchLoop waits for a slot here (back-pressure) instead of being
// refused with SS_ERR_STATE and falling back to synchronous calls.  The host's plan for batch i+1 and the copy-out of batch i-1
// then run under the kernels of batch i (INTEGRATION.md §4).
type inflight struct {
	snap   *deviceIndex
	ticket spaghetti.Ticket
	batch  []*request
}

var flights = make(chan inflight, spaghetti.ScoreInflight)

// slots: one token per batch between Submit and the end of its Collect (see above)
var slots = make(chan struct{}, spaghetti.ScoreInflight)

// replyAll hands every request of a collected batch its hits; a panic on the way still leaves every waiter with a reply.
func replyAll(batch []*request, hits [][]spaghetti.Hit) {
	for i, r := range batch {
		r.reply <- reply{hits[i], nil}
	}
}

func collectLoop() {
	for f := range flights {
		func() {
			answered := false
			defer func() {
				if p := recover(); p != nil && !answered {
					for _, r := range f.batch {
						select {
						case r.reply <- reply{nil, fmt.Errorf("retrieval batch failed: %v", p)}:
						default: // this caller already has its reply
						}
					}
				}
			}()
			released := false
			release := func() { // the ticket is spent whatever Collect did (a panic included): the next Submit may go ahead
				if !released {
					released = true
					<-slots
				}
			}
			defer release()
			hits, err := f.snap.scorer.Collect(f.ticket)
			release()
			if err == nil {
				replyAll(f.batch, hits)
				answered = true
				return
			}
			serve(f.snap, f.batch) // the batch failed as a whole: one synchronous call per request
			answered = true
		}()
	}
}

// serve answers one batch; every waiter gets a reply or an error, whatever happens (a panic below the library
// boundary included), and the batcher goroutine survives.
func serve(snap *deviceIndex, batch []*request) {
	answered := 0
	defer func() {
		if p := recover(); p != nil {
			for _, r := range batch[answered:] {
				r.reply <- reply{nil, fmt.Errorf("retrieval batch failed: %v", p)}
			}
		}
	}()
	hits, err := score(snap, batch)
	if err == nil {
		for i, r := range batch {
			r.reply <- reply{hits[i], nil}
			answered = i + 1
		}
		return
	}
	// The library refuses a batch as a whole (limits are validated per request before enqueueing, so this is the
	// unexpected case): run the requests one by one, so that only the offending request sees the error.
	for i, r := range batch {
		one, e := score(snap, batch[i:i+1])
		if e != nil {
			r.reply <- reply{nil, e}
		} else {
			r.reply <- reply{one[0], nil}
		}
		answered = i + 1
	}
}

func batchLoop() {
	for first := range reqs {
		batch := []*request{first}
		timer := time.NewTimer(batchWindow)
	collect:
		for len(batch) < maxBatch {
			select {
			case r := <-reqs:
				batch = append(batch, r)
			case <-timer.C:
				break collect
			}
		}
		timer.Stop()
		// requests tokenised against different snapshots (a Refresh fell into the window) are served per snapshot
		for len(batch) > 0 {
			snap := batch[0].snap
			var same, rest []*request
			for _, r := range batch {
				if r.snap == snap {
					same = append(same, r)
				} else {
					rest = append(rest, r)
				}
			}
			// enqueue and move on; if the library refuses the batch as a whole, serve it request by request right here
			qPtr, qTerms, pPtr, pTerms, qLen := concat(same)
			slots <- struct{}{} // waits while ScoreInflight batches are between Submit and Collect
			if t, err := snap.scorer.Submit(qPtr, qTerms, pPtr, pTerms, qLen, nil, topK); err == nil {
				flights <- inflight{snap, t, same} // never blocks: the channel holds as many batches as there are slots
			} else {
				<-slots
				serve(snap, same)
			}
			batch = rest
		}
	}
}

// Refresh retires the device copy of the tables; the next Retrieve flattens and uploads them again (call after a
// re-crawl has rewritten inv[*]/forw[3..4]: the reference re-reads BadgerDB on every request and needs no such call).
// Requests already running finish on the snapshot they started with; it is destroyed — scorer, then both tables — when
// the last of them is done.
func Refresh() {
	devMu.Lock()
	old := dev
	dev = nil
	devMu.Unlock()
	if old != nil {
		go old.close()
	}
}

// acquire returns the current snapshot with this request registered on it (release with snap.users.Done()).
func acquire(ctx context.Context, forw []db.DB, inv []db.DB) *deviceIndex {
	devMu.RLock()
	if d := dev; d != nil {
		d.users.Add(1)
		devMu.RUnlock()
		return d
	}
	devMu.RUnlock()
	devMu.Lock()
	defer devMu.Unlock()
	if dev == nil {
		dev = load(ctx, forw, inv)
	}
	dev.users.Add(1)
	return dev
}

func Retrieve(query string, ctx context.Context, forw []db.DB, inv []db.DB) []Rank_combined {
	snap := acquire(ctx, forw, inv)
	defer snap.users.Done()
	batchers.Do(func() { go batchLoop(); go collectLoop() })

	// main_retrieve.go:17-36 — query parsing, unchanged
	phrases := getPhrase(query)
	for _, term := range phrases {
		query = strings.Replace(query, "\""+string(term)+"\"", "", 1)
	}
	queryTokenised := parser.Laundry(strings.Join(strings.Fields(query), " "))
	phraseTokenised := parser.Laundry(strings.Join(phrases, " "))

	toIDs := func(tokens []string) []uint32 {
		ids := make([]uint32, len(tokens))
		for i, tok := range tokens {
			sum := md5.Sum([]byte(tok))
			if id, ok := snap.termID[hex.EncodeToString(sum[:])]; ok {
				ids[i] = id
			} else {
				ids[i] = 0xFFFFFFFF // badger.ErrKeyNotFound: no postings (main_retrieve.go:193,218)
			}
		}
		return ids
	}
	// all quoted phrases form ONE phrase (main_retrieve.go:26); it is matched on the device from the
	// positional part of the postings (retrieval/phrase.go -> ss_score_topk_phrase)
	r := &request{snap: snap, qTerms: toIDs(queryTokenised), pTerms: toIDs(phraseTokenised),
		qLen:  int32(len(queryTokenised) + len(phraseTokenised)), // main_retrieve.go:90
		reply: make(chan reply, 1)}
	// The library's per-request limits are checked HERE, in the request's own goroutine (net/http recovers a panic of a
	// handler goroutine and the other requests of the batch never see it); the reference has no such limits.
	if len(r.pTerms) > spaghetti.MaxPhraseTerms {
		panic(fmt.Errorf("quoted phrase of %d words: at most %d are supported", len(r.pTerms), spaghetti.MaxPhraseTerms))
	}
	distinct := make(map[uint32]struct{}, len(r.qTerms))
	for _, t := range r.qTerms {
		distinct[t] = struct{}{}
	}
	if len(distinct) > spaghetti.MaxQueryTerms {
		panic(fmt.Errorf("query of %d distinct words: at most %d are supported", len(distinct), spaghetti.MaxQueryTerms))
	}
	reqs <- r
	ans := <-r.reply
	if ans.err != nil {
		panic(ans.err) // error policy of the reference: panic in the request goroutine
	}
	hits := ans.hits

	out := make([]Rank_combined, 0, len(hits))
	for _, h := range hits {
		docHash := snap.docName[h.Doc]
		meta := <-getDocInfo(ctx, docHash, forw) // get_metadata.go:211-235, for the winners only
		meta.PageRank = h.PageRank               // get_metadata.go:68
		meta.FinalRank = h.Final                 // get_metadata.go:69
		meta.Summary = <-getSummary(docHash, query, phrases)
		out = append(out, meta)
	}
	return out
}
