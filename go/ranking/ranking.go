// Package ranking — drop-in for the reference's package of the same name.
//
// Same exported signatures as ranking/pagerank.go:14 and ranking/term_weighting.go:10; the
// bodies flatten the BadgerDB tables to dense-id arrays, make ONE call into the HIP library per
// function (not per iteration / per term) and write the results back in the reference's
// table formats (forw[3]: map[category]float64, forw[4]: map{"title","body"}float64,
// inv[*]: map[docHash][]float32 with listPos[0] replaced).
//
// Call order is the reference's (cmd/crawl/start_crawl.go:175-177): PageRank first, then
// title, then body — UpdateTermWeights reads len(forw[3]) (term_weighting.go:13-17, Q7).
package ranking

import (
	"context"
	"encoding/json"
	"log"
	"sort"

	db "github.com/nwihardjo/SpaghettiSearch/database"

	"github.com/nwihardjo/SpaghettiSearch/go/spaghetti"
)

// denseIDs assigns 0..n-1 to md5-hex hashes in sorted order (deterministic, unlike Go map order).
type denseIDs struct {
	id   map[string]uint32
	name []string
}

// TwoVectors switches UpdateTopicSensitivePagerank to the library's two-vector form (see updatePagerank).
var TwoVectors = false

func newDenseIDs(keys map[string]struct{}) *denseIDs {
	d := &denseIDs{id: make(map[string]uint32, len(keys)), name: make([]string, 0, len(keys))}
	for k := range keys {
		d.name = append(d.name, k)
	}
	sort.Strings(d.name)
	for i, k := range d.name {
		d.id[k] = uint32(i)
	}
	return d
}

// UpdateTopicSensitivePagerank keeps the reference's signature and behaviour (every topic teleports uniformly).
func UpdateTopicSensitivePagerank(ctx context.Context, dampingFactor float64, convergenceCriterion float64, forward []db.DB) {
	updatePagerank(ctx, dampingFactor, convergenceCriterion, forward, nil)
}

// UpdateTopicSensitivePagerankTeleport is the OPT-IN form (SURVEY.md §8f-3; no counterpart in the reference, which
// advertises topic-sensitive PageRank, README.md:9, but differs per topic by the start value only): every category
// teleports into the pages TopicTeleportSets assigns to it.  inverted = inv (inv[0], inv[1], inv[2] are read).
func UpdateTopicSensitivePagerankTeleport(ctx context.Context, dampingFactor float64, convergenceCriterion float64, forward []db.DB, inverted []db.DB) {
	updatePagerank(ctx, dampingFactor, convergenceCriterion, forward, TopicTeleportSets(ctx, forward, inverted))
}

// TopicTeleportSets: category -> doc hashes.  A page joins the set of the category whose ODP keywords
// (inv[2][wordHash] = map[category]frequency, forw[5][category]["wordCount"]; crawler/ODP-scraper.go:97-139) it holds
// most of, by the estimate computeTopicProbs uses for a query (tf / wordCount, main_retrieve.go:143-145):
//     mass(page, c) = sum over the words w of the page (inv[0] or inv[1] row of w holds the page) of inv[2][w][c] / wordCount(c)
// ties go to the first category in key order; a page without keyword hits joins no set.
func TopicTeleportSets(ctx context.Context, forward []db.DB, inverted []db.DB) map[string][]string {
	catComp, err := forward[5].Iterate(ctx)
	if err != nil {
		panic(err)
	}
	cats := make([]string, 0, len(catComp.KV))
	wordCount := make(map[string]float64, len(catComp.KV))
	for _, kv := range catComp.KV {
		val := make(map[string]float64, 2)
		if err = json.Unmarshal(kv.Value, &val); err != nil {
			panic(err)
		}
		cats = append(cats, string(kv.Key))
		wordCount[string(kv.Key)] = val["wordCount"]
	}
	sort.Strings(cats)
	kwComp, err := inverted[2].Iterate(ctx)
	if err != nil {
		panic(err)
	}
	mass := make(map[string][]float64)
	for _, kv := range kwComp.KV {
		var freq map[string]float64
		if err = json.Unmarshal(kv.Value, &freq); err != nil {
			panic(err)
		}
		share := make([]float64, len(cats))
		hit := false
		for i, c := range cats {
			if f, ok := freq[c]; ok && wordCount[c] > 0 {
				share[i] = f / wordCount[c]
				hit = true
			}
		}
		if !hit {
			continue
		}
		for t := 0; t < 2; t++ {
			v, err := inverted[t].Get(ctx, string(kv.Key))
			if err != nil {
				continue // the keyword is in no page of this table
			}
			for docHash := range v.(map[string][]float32) {
				m, ok := mass[docHash]
				if !ok {
					m = make([]float64, len(cats))
					mass[docHash] = m
				}
				for i := range cats {
					m[i] += share[i]
				}
			}
		}
	}
	sets := make(map[string][]string, len(cats))
	for _, c := range cats {
		sets[c] = nil
	}
	for docHash, m := range mass {
		best := 0
		for i := 1; i < len(cats); i++ {
			if m[i] > m[best] {
				best = i
			}
		}
		if m[best] > 0 {
			sets[cats[best]] = append(sets[cats[best]], docHash)
		}
	}
	for _, c := range cats {
		sort.Strings(sets[c])
	}
	return sets
}

func updatePagerank(ctx context.Context, dampingFactor float64, convergenceCriterion float64, forward []db.DB, teleport map[string][]string) {
	log.Printf("Ranking with damping factor='%f', convergence_criteria='%f'", dampingFactor, convergenceCriterion)

	// pagerank.go:17-44 — node set = parents U children (frontier pages are nodes without children)
	nodesCompressed, err := forward[2].Iterate(ctx)
	if err != nil {
		panic(err)
	}
	all := make(map[string]struct{})
	children := make(map[string][]string, len(nodesCompressed.KV))
	for _, kv := range nodesCompressed.KV {
		var c []string
		if err = json.Unmarshal(kv.Value, &c); err != nil {
			panic(err)
		}
		for _, h := range c {
			all[h] = struct{}{}
		}
		children[string(kv.Key)] = c
		all[string(kv.Key)] = struct{}{}
	}
	ids := newDenseIDs(all)
	n := len(ids.name)

	// out-edge CSR over dense ids
	outPtr := make([]uint64, n+1)
	for p, c := range children {
		outPtr[ids.id[p]+1] = uint64(len(c))
	}
	for i := 0; i < n; i++ {
		outPtr[i+1] += outPtr[i]
	}
	outDst := make([]uint32, outPtr[n])
	for p, c := range children {
		base := outPtr[ids.id[p]]
		for j, h := range c {
			outDst[base+uint64(j)] = ids.id[h]
		}
	}

	// pagerank.go:46-63 — one power iteration per category, differing by numPages only
	categoryCompressed, err := forward[5].Iterate(ctx)
	if err != nil {
		panic(err)
	}
	cats := make([]string, 0, len(categoryCompressed.KV))
	nTopic := make([]int32, 0, len(categoryCompressed.KV))
	for _, kv := range categoryCompressed.KV {
		val := make(map[string]float64, 2)
		if err = json.Unmarshal(kv.Value, &val); err != nil {
			panic(err)
		}
		log.Printf("number of webnodes in %s is %d", string(kv.Key), int(val["numPages"]))
		cats = append(cats, string(kv.Key))
		nTopic = append(nTopic, int32(int(val["numPages"])))
	}

	bw := forward[3].BatchWrite_init(ctx)
	defer bw.Cancel(ctx)
	if job := spaghetti.Job(); job.World > 1 {
		if teleport != nil {
			panic("teleport sets with several GPUs: use the step-wise calls (ss_pr_set_teleport takes the full sets on every rank)")
		}
		// Several GPUs (one crawl process per GPU, SS_RANK/SS_WORLD): every process flattens the same tables to the same
		// ids (sorted hashes), keeps the destination rows of ITS doc-range shard, runs its part of the sweep with one
		// RCCL all-gather per iteration inside the library, and writes its own rows of forw[3].  K > 16 in groups of 16.
		g := spaghetti.Default().NewGraphShard(outPtr, outDst, job.Rank, job.World)
		defer g.Close()
		var rows []uint32
		ranks := make([][]float64, 0, (len(nTopic)+15)/16)
		for k0 := 0; k0 < len(nTopic); k0 += 16 {
			k1 := k0 + 16
			if k1 > len(nTopic) {
				k1 = len(nTopic)
			}
			r, rk, _ := g.PageRankSharded(dampingFactor, convergenceCriterion, nTopic[k0:k1])
			rows = r
			ranks = append(ranks, rk)
		}
		for i, v := range rows {
			PR := make(map[string]float64, len(cats))
			for k, c := range cats {
				blk := ranks[k/16]
				PR[c] = blk[(k%16)*len(rows)+i]
			}
			if err := bw.BatchSet(ctx, ids.name[v], PR); err != nil {
				panic(err)
			}
		}
	} else {
		g := spaghetti.Default().NewGraph(outPtr, outDst)
		defer g.Close()
		var rank []float64
		if teleport == nil {
			// TwoVectors (opt-in, off by default): every category's ranks from two vectors — the categories differ only in their
			// start value 1/numPages and the recurrence maps a ratio of affine forms in it onto itself (library option "pr.affine";
			// same ranks to ~1e-15, not the reference's operation order; the cost no longer grows with the category count)
			if TwoVectors {
				spaghetti.Default().SetOption("pr.affine", 1)
				defer spaghetti.Default().SetOption("pr.affine", spaghetti.OptionDefault)
			}
			rank, _ = g.PageRank(dampingFactor, convergenceCriterion, nTopic) // all categories, device-resident loop
		} else {
			sets := make([][]uint32, len(cats))
			for k, c := range cats {
				for _, h := range teleport[c] {
					if v, ok := ids.id[h]; ok { // a page outside the link graph has no rank to receive
						sets[k] = append(sets[k], v)
					}
				}
			}
			rank, _ = g.PageRankTeleport(dampingFactor, convergenceCriterion, nTopic, sets)
		}

		// pagerank.go:65-82 — forw[3][doc] = map[category]rank
		for v, name := range ids.name {
			PR := make(map[string]float64, len(cats))
			for k, c := range cats {
				PR[c] = rank[k*n+v]
			}
			if err := bw.BatchSet(ctx, name, PR); err != nil {
				panic(err)
			}
		}
	}
	if err = bw.Flush(ctx); err != nil {
		panic(err)
	}
}

func UpdateTermWeights(ctx context.Context, inv *db.DB, forw []db.DB, info string) {
	// term_weighting.go:12-17 — N = number of PageRank nodes
	nodes, err := forw[3].Iterate(ctx)
	if err != nil {
		panic(err)
	}
	totalDocs := uint64(len(nodes.KV))
	docSet := make(map[string]struct{}, len(nodes.KV))
	for _, kv := range nodes.KV {
		docSet[string(kv.Key)] = struct{}{}
	}

	comp, err := (*inv).Iterate(ctx)
	if err != nil {
		panic(err)
	}
	// decode every row once; keep the positional tails for the write-back
	rows := make([]map[string][]float32, len(comp.KV))
	for i := range comp.KV {
		if err = json.Unmarshal(comp.KV[i].Value, &rows[i]); err != nil {
			panic(err)
		}
		for h := range rows[i] {
			docSet[h] = struct{}{}
		}
	}
	docs := newDenseIDs(docSet)

	// term-major CSR, postings sorted by dense doc id
	termPtr := make([]uint64, len(rows)+1)
	for i, r := range rows {
		termPtr[i+1] = termPtr[i] + uint64(len(r))
	}
	postDoc := make([]uint32, termPtr[len(rows)])
	postTf := make([]float32, termPtr[len(rows)])
	for i, r := range rows {
		seg := postDoc[termPtr[i]:termPtr[i+1]]
		j := 0
		for h := range r {
			seg[j] = docs.id[h]
			j++
		}
		sort.Slice(seg, func(a, b int) bool { return seg[a] < seg[b] })
		for j, d := range seg {
			postTf[termPtr[i]+uint64(j)] = r[docs.name[d]][0] // listPos[0] = normalised tf
		}
	}

	ix := spaghetti.Default().NewIndex(uint64(len(docs.name)), termPtr, postDoc, postTf)
	defer ix.Close()
	w, mag := ix.TfIdfBuild(totalDocs) // idf, w = tf*idf, mag = sqrt(sum w^2)   (term_weighting.go:37-44,72)

	// term_weighting.go:42,47 — write the weights back in place
	bw := (*inv).BatchWrite_init(ctx)
	defer bw.Cancel(ctx)
	for i, r := range rows {
		for j := termPtr[i]; j < termPtr[i+1]; j++ {
			r[docs.name[postDoc[j]]][0] = w[j]
		}
		if err = bw.BatchSet(ctx, string(comp.KV[i].Key), r); err != nil {
			panic(err)
		}
	}
	if err = bw.Flush(ctx); err != nil {
		panic(err)
	}

	// term_weighting.go:59-123 (saveMagnitude): only docs that occur in this table get a value
	pageMagnitude := make(map[string]float64)
	for _, d := range postDoc {
		pageMagnitude[docs.name[d]] = mag[d]
	}
	saveMagnitude(ctx, pageMagnitude, &forw[4], info)
}

// saveMagnitude merges the (already square-rooted) magnitudes into forw[4][doc][info], keeping the
// other key — same table semantics as term_weighting.go:59-123.
func saveMagnitude(ctx context.Context, pageMagnitude map[string]float64, forw *db.DB, info string) {
	comp, err := (*forw).Iterate(ctx)
	if err != nil {
		panic(err)
	}
	bw := (*forw).BatchWrite_init(ctx)
	defer bw.Cancel(ctx)
	for i := range comp.KV {
		key := string(comp.KV[i].Key)
		var v map[string]float64
		if err = json.Unmarshal(comp.KV[i].Value, &v); err != nil {
			panic(err)
		}
		v[info] = pageMagnitude[key] // missing => 0, like math.Sqrt(0) at term_weighting.go:97
		delete(pageMagnitude, key)
		if err = bw.BatchSet(ctx, key, v); err != nil {
			panic(err)
		}
	}
	for h, m := range pageMagnitude {
		if err = bw.BatchSet(ctx, h, map[string]float64{info: m}); err != nil {
			panic(err)
		}
	}
	if err = bw.Flush(ctx); err != nil {
		panic(err)
	}
}
