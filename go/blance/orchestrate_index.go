// SURVEY.md 8(f-4): the orchestrator's move selection at a million partitions.  Host-side Go by the north
// star -- no device code.  ADD to package blance; the three call-site changes are listed below.  Like the
// rest of go/blance it cannot be compiled in this repository's build image.
//
// findAvailableMovesUnlocked (orchestrate.go:749-763) rebuilds, every supply round, a map node -> the
// partitions whose NEXT move goes to that node, by walking all of o.mapPartitionToNextMoves: O(P) per
// round, and the number of rounds grows with P, so a rebalance of a million partitions is quadratic
// there once the planner in front of it takes milliseconds.  The index below keeps the same map
// incrementally: a partition sits in the bucket of the node of its next move and changes bucket only when
// its Next advances (orchestrate.go:689) -- O(1) per completed move, nothing per round.
//
// Call sites (all under o.m, like the fields they replace):
//   * OrchestrateMoves, after mapPartitionToNextMoves is filled (orchestrate.go:271-290):
//         o.moveIndex = newMoveIndex(mapPartitionToNextMoves)
//   * runSupplyMoves, instead of o.findAvailableMovesUnlocked() (orchestrate.go:521):
//         availableMoves := o.moveIndex.available()
//   * where a move completes, next to nextMoves[i].Next++ (orchestrate.go:689):
//         o.moveIndex.advanced(nextMoves[i], oldNode)      // oldNode = Moves[Next-1].Node
// The order of the partitions inside a bucket is not specified by the reference (it ranges over a Go map);
// findMove (orchestrate.go:158-177) picks by weight, not by position.

package blance

type moveIndex struct {
	buckets map[string]map[*NextMoves]struct{} // node -> partitions whose next move targets it
}

func newMoveIndex(all map[string]*NextMoves) *moveIndex {
	ix := &moveIndex{buckets: map[string]map[*NextMoves]struct{}{}}
	for _, nm := range all {
		ix.insert(nm)
	}
	return ix
}

func (ix *moveIndex) insert(nm *NextMoves) {
	if nm.Next >= len(nm.Moves) {
		return // nothing left to do for this partition
	}
	node := nm.Moves[nm.Next].Node
	b := ix.buckets[node]
	if b == nil {
		b = map[*NextMoves]struct{}{}
		ix.buckets[node] = b
	}
	b[nm] = struct{}{}
}

// advanced re-files a partition whose Next was just incremented; oldNode is the node of the move that
// completed (its bucket held the partition until now).
func (ix *moveIndex) advanced(nm *NextMoves, oldNode string) {
	if b := ix.buckets[oldNode]; b != nil {
		delete(b, nm)
		if len(b) == 0 {
			delete(ix.buckets, oldNode)
		}
	}
	ix.insert(nm)
}

// available is findAvailableMovesUnlocked's result: keyed by node name, the partitions with a next move for it.
func (ix *moveIndex) available() map[string][]*NextMoves {
	out := make(map[string][]*NextMoves, len(ix.buckets))
	for node, b := range ix.buckets {
		lst := make([]*NextMoves, 0, len(b))
		for nm := range b {
			lst = append(lst, nm)
		}
		out[node] = lst
	}
	return out
}
