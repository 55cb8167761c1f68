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

// Geometry of an image request: falsy fields take the API defaults (lib/image.js:3-9 of the reference: width / height 0,
// pixel 'rgba', rows padded to 4 bytes), then the four checks in the reference's order with its messages (lib/image.js:10-20).
function resolveLayout(req) {
	const layout = {
		width: req.width ? req.width : 0,
		height: req.height ? req.height : 0,
		pixel: req.pixel ? req.pixel : 'rgba',
	};
	const bpp = sizeOf(layout.pixel);
	const payload = layout.width * bpp;          // bytes of a row that carry pixels
	layout.stride = req.stride ? req.stride : ((payload + 3) & ~3);
	if (bpp === 0) throw new Error("invalid pixel format " + layout.pixel);
	if (layout.stride < payload) throw new Error("stride too short");
	if (layout.width < 0 || layout.height < 0) throw new Error("invalid dimensions");
	layout.bytes = layout.stride * layout.height;                         // a freshly allocated image
	layout.minBytes = layout.stride * (layout.height - 1) + payload;      // the last row may stop at its payload
	return layout;
}

class Image {
	constructor(opt) {
		const req = opt || {};
		this.data = req.data;                  // first, so the property order matches the reference's objects
		const layout = resolveLayout(req);
		this.width = layout.width;
		this.height = layout.height;
		this.pixel = layout.pixel;
		this.stride = layout.stride;
		let store = req.data;
		if (!store && layout.bytes !== 0) store = Buffer.alloc(layout.bytes);   // zero-filled, like the reference
		if (store && store.length < layout.minBytes) throw new Error("image data too small");
		this.data = store;
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

	// byte offset of pixel (x, y) inside this.data
	offsetOf(x, y) { return y * this.stride + x * this.pixelSize(); }

	// a w x h window with its top-left corner at (x, y): shares this image's memory and stride; the view's buffer ends
	// with the payload of its last row (lib/image.js:76-87)
	subView(x, y, w, h) {
		const first = this.offsetOf(x, y);
		const last = this.offsetOf(x + w, y + h - 1);   // one past the last payload byte of the window
		return new Image({ data: this.data.slice(first, last), stride: this.stride, pixel: this.pixel, width: w, height: h });
	}

	// copies the common top-left rectangle into `target`, payload bytes only (lib/image.js:89-96)
	copy(target) {
		if (target.pixel != this.pixel) throw new Error("can't copy pixels between different pixel types");
		const rows = Math.min(this.height, target.height);
		const bytes = Math.min(this.width, target.width) * this.pixelSize();
		for (let from = 0, to = 0, y = 0; y < rows; ++y, from += this.stride, to += target.stride)
			this.data.copy(target.data, to, from, from + bytes);
	}
}

Image.bufferCompare = Buffer.compare || function (a, b) {
	const n = Math.min(a.length, b.length);
	for (let i = 0; i < n; ++i) if (a[i] !== b[i]) return a[i] < b[i] ? -1 : 1;
	return a.length === b.length ? 0 : (a.length < b.length ? -1 : 1);
};

exports.Image = Image;
