package net.preibisch.simulation.gpu;

import java.util.Random;

/**
 * java.util.Random with its 48-bit state in plain sight.  The JDK specifies every draw of Random in terms of
 * next(int), so overriding next() with the specified LCG reproduces `new Random( seed )` bit for bit while letting
 * the native phantom generator (mvsim_draw_spheres) continue the very same stream: the state goes down as a long
 * and comes back advanced by exactly the draws the reference's drawSpheres would have made.
 *
 * SimulateMultiViewDataset keeps `public static Random rnd = new Random( 464232194 )` (:76); the one-line change
 * for the GPU path is `rnd = new GpuRandom( 464232194 )`.
 */
public class GpuRandom extends Random
{
	private static final long serialVersionUID = 1L;
	private static final long MULT = 0x5DEECE66DL, ADD = 0xBL, MASK = ( 1L << 48 ) - 1;

	private long state;

	public GpuRandom( final long seed )
	{
		super( seed );          // calls setSeed below
	}

	@Override
	public synchronized void setSeed( final long seed )
	{
		state = ( seed ^ MULT ) & MASK;
	}

	@Override
	protected synchronized int next( final int bits )
	{
		state = ( state * MULT + ADD ) & MASK;
		return ( int ) ( state >>> ( 48 - bits ) );
	}

	/** the raw 48-bit LCG state (what mvsim_draw_spheres takes and returns) */
	public synchronized long getState() { return state; }

	public synchronized void setState( final long s ) { state = s & MASK; }
}
