"use strict";
// Image — the raw pixel container of the picha API: {width, height, stride, pixel, data}.
// Own implementation of the behaviour of the reference's lib/image.js (constructor defaults and
// error strings lib/image.js:3-21, helpers :34-96).  One deliberate difference: the reference's
// size table spells 'r16g16' as 'r16b16' (lib/image.js:31), so it rejects r16g16 images; this
// table has the right key.

const BYTES_PER_PIXEL = Object.freeze({
	rgb: 3, rgba: 4, grey: 1, greya: 2, r16: 2, r16g16: 4, r16g16b16: 6, r16g16b16a16: 8,
});

function sizeOf(pixel) {
	return Object.prototype.hasOwnProperty.call(BYTES_PER_PIXEL, pixel) ? BYTES_PER_PIXEL[pixel] : 0;
}

class Image {
	constructor(opt) {
		opt = opt || {};
		this.data = opt.data;
		this.width = opt.width || 0;
		this.height = opt.height || 0;
		this.pixel = opt.pixel || 'rgba';
		const psize = sizeOf(this.pixel);
		this.stride = opt.stride || ((this.width * psize + 3) & ~3);
		if (psize === 0) throw new Error("invalid pixel format " + this.pixel);
		if (this.stride < this.width * psize) throw new Error("stride too short");
		if (this.width < 0 || this.height < 0) throw new Error("invalid dimensions");
		if (this.stride * this.height !== 0 && !this.data) this.data = Buffer.alloc(this.stride * this.height);
		if (this.data && this.data.length < this.stride * (this.height - 1) + this.width * psize)
			throw new Error("image data too small");
	}

	static pixelSize(pixel) { return sizeOf(pixel); }

	pixelSize() { return sizeOf(this.pixel); }

	// payload bytes of row y (no padding)
	row(y) {
		const start = y * this.stride;
		return this.data.slice(start, start + this.width * this.pixelSize());
	}

	sameShape(o) {
		return this.width === o.width && this.height == o.height && this.pixel == o.pixel;
	}

	// payload comparison only: row padding is never looked at
	equalPixels(o) {
		if (!this.sameShape(o)) return false;
		for (let y = 0; y < this.height; ++y)
			if (Image.bufferCompare(this.row(y), o.row(y)) !== 0) return false;
		return true;
	}

	// mean absolute byte difference over the payload; 255 for images of different shape
	avgChannelDiff(o) {
		if (!this.sameShape(o)) return 255;
		const rw = this.width * this.pixelSize();
		let sum = 0;
		for (let y = 0; y < this.height; ++y) {
			const a = this.row(y), b = o.row(y);
			for (let x = 0; x < rw; ++x) sum += Math.abs(a[x] - b[x]);
		}
		return sum / (this.height * rw);
	}

	// a window onto the same memory (same stride)
	subView(x, y, w, h) {
		const p = this.pixelSize();
		const off = y * this.stride + x * p;
		const len = (h - 1) * this.stride + w * p;
		return new Image({ width: w, height: h, pixel: this.pixel, stride: this.stride, data: this.data.slice(off, off + len) });
	}

	copy(target) {
		if (target.pixel != this.pixel) throw new Error("can't copy pixels between different pixel types");
		const rw = this.pixelSize() * Math.min(this.width, target.width);
		const h = Math.min(this.height, target.height);
		for (let y = 0; y < h; ++y)
			this.data.copy(target.data, y * target.stride, y * this.stride, y * this.stride + rw);
	}
}

Image.bufferCompare = Buffer.compare || function (a, b) {
	const n = Math.min(a.length, b.length);
	for (let i = 0; i < n; ++i) if (a[i] !== b[i]) return a[i] < b[i] ? -1 : 1;
	return a.length === b.length ? 0 : (a.length < b.length ? -1 : 1);
};

exports.Image = Image;
