// SURVEY.md 8(f-4): the orchestrator's move selection at a million partitions.  Host-side Go by the north
// star -- no device code.  ADD to package blance; the three call-site changes are listed below.  Like the
// rest of go/blance it cannot be compiled in this repository's build image; its compiled twin, structure for
// structure, is blance_amd/csrc/host/move_index.hpp, checked against the reference's rescan by a randomised
// simulation of the supply rounds (move_index_sim.cpp, tests/test_move_index.py).
//
// findAvailableMovesUnlocked (orchestrate.go:749-763) rebuilds, every supply round, a map node -> the
// partitions whose NEXT move goes to that node, by walking all of o.mapPartitionToNextMoves: O(P) per
// round; filterNextPlausibleMovesForNode (orchestrate.go:482-504) then materialises a node's whole bucket
// once per slot for FindMoveFunc.  moveIndex keeps the buckets incrementally instead:
//   * a partition sits in bucket (node, op) of its next move; position tables make removal a swap -- O(1)
//     per completed move (orchestrate.go:689), nothing per round;
//   * nodes with pending moves are a dense list: a round touches those nodes only;
//   * lowestWeight(node, count) answers filterNextPlausibleMovesForNode for the default FindMoveFunc
//     LowestWeightPartitionMoveForNode (orchestrate.go:174-184): `count` moves in ascending MoveOpWeight.  The
//     order among equal weights is unspecified in the reference (it ranges over a Go map, orchestrate.go:755);
//   * bucket(node) lists a node's pending partitions for an application FindMoveFunc.
//
// LOCKING.  Every method of moveIndex reads or writes the live buckets and must run with o.m HELD -- like the
// fields it replaces.  Nothing of the index may be touched after o.m.Unlock(): the reference's supply loop
// iterates AFTER unlocking (orchestrate.go:521-560) while runSupplyMove goroutines advance Next under o.m
// (orchestrate.go:684-691), and advanced() swap-removes -- it reorders `active` and the buckets.  So the loop
// works on a SNAPSHOT taken under the lock, exactly as the reference iterates the fresh map that
// findAvailableMovesUnlocked built under the lock: snapshot() returns fresh slices that the index never
// touches again.
//
// Call sites:
//   * OrchestrateMoves, after mapPartitionToNextMoves is filled (orchestrate.go:271-290), under o.m:
//         o.moveIndex = newMoveIndex(mapPartitionToNextMoves)
//   * runSupplyMoves, instead of o.findAvailableMovesUnlocked() (orchestrate.go:521), still under o.m:
//         avail := o.moveIndex.snapshot(count)        // node -> its `count` lowest-weight pending moves
//         o.m.Unlock()
//         for node, nxt := range avail { ... }         // replaces the per-node filter call (orchestrate.go:549)
//     (with an application FindMoveFunc: o.moveIndex.snapshotBuckets(), then filterNextPlausibleMovesForNode)
//   * where a move completes, next to nextMoves[i].Next++ (orchestrate.go:689), under o.m:
//         o.moveIndex.advanced(nextMoves[i])

package blance

// op classes in ascending MoveOpWeight (orchestrate.go:187-192).  An op the table does not know has weight 0 in
// the reference (a missing map key) and therefore ranks BEFORE promote: class 0; the four known ops follow.
const (
	moveOpUnknown = 0
	moveOpClasses = 5
)

var moveOpClass = map[string]int{"promote": 1, "demote": 2, "add": 3, "del": 4}

func opClass(op string) int {
	if c, ok := moveOpClass[op]; ok {
		return c
	}
	return moveOpUnknown
}

type moveIndexNode struct {
	byOp      [moveOpClasses][]*NextMoves // partitions whose next move is (this node, op class)
	pending   int
	activePos int // position in moveIndex.active, -1 if pending == 0
}

type moveIndex struct {
	nodes  map[string]*moveIndexNode
	active []string           // nodes with pending > 0
	where  map[*NextMoves]int // position inside its bucket
	total  int
}

func newMoveIndex(all map[string]*NextMoves) *moveIndex {
	ix := &moveIndex{nodes: map[string]*moveIndexNode{}, where: make(map[*NextMoves]int, len(all))}
	for _, nm := range all {
		ix.insert(nm)
	}
	return ix
}

func (ix *moveIndex) insert(nm *NextMoves) {
	if nm.Next >= len(nm.Moves) {
		return // nothing left to do for this partition
	}
	m := nm.Moves[nm.Next]
	n := ix.nodes[m.Node]
	if n == nil {
		n = &moveIndexNode{activePos: -1}
		ix.nodes[m.Node] = n
	}
	op := opClass(m.Op)
	ix.where[nm] = len(n.byOp[op])
	n.byOp[op] = append(n.byOp[op], nm)
	if n.pending == 0 {
		n.activePos = len(ix.active)
		ix.active = append(ix.active, m.Node)
	}
	n.pending++
	ix.total++
}

func (ix *moveIndex) remove(nm *NextMoves, node string, opName string) {
	n := ix.nodes[node]
	op := opClass(opName)
	at, ok := ix.where[nm]
	if n == nil || !ok || at >= len(n.byOp[op]) || n.byOp[op][at] != nm {
		return // not filed (it had no move left when the index was built): nothing to take out
	}
	b := n.byOp[op]
	last := b[len(b)-1]
	b[at] = last
	ix.where[last] = at
	b[len(b)-1] = nil
	n.byOp[op] = b[:len(b)-1]
	delete(ix.where, nm)
	n.pending--
	if n.pending == 0 {
		ln := ix.active[len(ix.active)-1]
		ix.active[n.activePos] = ln
		ix.nodes[ln].activePos = n.activePos
		ix.active = ix.active[:len(ix.active)-1]
		n.activePos = -1
	}
	ix.total--
}

// advanced re-files a partition whose Next was just incremented (orchestrate.go:689).
func (ix *moveIndex) advanced(nm *NextMoves) {
	old := nm.Moves[nm.Next-1]
	ix.remove(nm, old.Node, old.Op)
	ix.insert(nm)
}

// snapshot: for every node with pending moves, up to count of them in ascending MoveOpWeight -- what the supply
// loop needs, as FRESH slices (call with o.m held; the result may be used after Unlock).
func (ix *moveIndex) snapshot(count int) map[string][]*NextMoves {
	out := make(map[string][]*NextMoves, len(ix.active))
	for _, node := range ix.active {
		out[node] = ix.lowestWeight(node, count)
	}
	return out
}

// snapshotBuckets: findAvailableMovesUnlocked()'s map itself, built from the index (fresh slices; o.m held).
func (ix *moveIndex) snapshotBuckets() map[string][]*NextMoves {
	out := make(map[string][]*NextMoves, len(ix.active))
	for _, node := range ix.active {
		out[node] = ix.bucket(node)
	}
	return out
}

// lowestWeight: up to count partitions whose next move goes to node, ascending MoveOpWeight (a fresh slice; o.m held).
func (ix *moveIndex) lowestWeight(node string, count int) []*NextMoves {
	n := ix.nodes[node]
	if n == nil {
		return nil
	}
	if count <= 0 {
		count = 1 // orchestrate.go:485-487
	}
	var out []*NextMoves
	for op := 0; op < moveOpClasses && count > 0; op++ {
		for i := 0; i < len(n.byOp[op]) && count > 0; i++ {
			out = append(out, n.byOp[op][i])
			count--
		}
	}
	return out
}

// bucket: every partition whose next move goes to node (findAvailableMovesUnlocked()[node]).
func (ix *moveIndex) bucket(node string) []*NextMoves {
	n := ix.nodes[node]
	if n == nil {
		return nil
	}
	out := make([]*NextMoves, 0, n.pending)
	for op := 0; op < moveOpClasses; op++ {
		out = append(out, n.byOp[op]...)
	}
	return out
}
